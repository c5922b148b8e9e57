"""Which broker / explorers receive the next weights.

Control-plane glue the learner process of the host framework expects on every Algorithm
(``alg.dist_model_policy.get_dist_info(...)`` / ``.add_processed_ctr_info(...)``, reference behaviour:
xt/algorithm/alg_utils.py:26-117).  A destination is ``{"broker_id": b, "explorer_id": ids}`` with -1 = everybody.
"""
import collections

EVERYBODY = -1


def _destination(broker=EVERYBODY, explorers=EVERYBODY):
    return {"broker_id": broker, "explorer_id": explorers}


def _restrict(dest, allowed):
    """Narrow a destination to the explorers in ``allowed`` (no-op without a restriction)."""
    if allowed:
        current = dest["explorer_id"]
        dest["explorer_id"] = allowed if current == EVERYBODY else [e for e in current if e in allowed]
    return dest


class DefaultAlgDistPolicy(object):
    """Broadcast to every explorer of every broker."""

    def __init__(self, actor_num, **kwargs):
        self.actor_num = actor_num
        self.default_policy = _destination()

    def add_processed_ctr_info(self, ctr_info):
        """Called by the learner for every consumed rollout message; only the data-driven policies care."""

    def get_dist_info(self, model_index, explorer_set=None):
        return _restrict(self.default_policy, explorer_set)


class DivideDistPolicy(DefaultAlgDistPolicy):
    """Model k goes to explorer k mod actor_num."""

    def get_dist_info(self, model_index, explorer_set=None):
        if model_index >= 0:
            self.default_policy["explorer_id"] = model_index % self.actor_num
        return _restrict(self.default_policy, explorer_set)


class FIFODistPolicy(DefaultAlgDistPolicy):
    """Answer exactly the explorers whose data went into the update (IMPALA-style asynchronous actors)."""

    def __init__(self, actor_num, prepare_times, **kwargs):
        super().__init__(actor_num, **kwargs)
        self.prepare_data_times = prepare_times
        self._pending = collections.deque()          # (broker_id, explorer_id, ...) of consumed messages

    def add_processed_ctr_info(self, ctr_info):
        self._pending.append(ctr_info)

    def get_dist_info(self, model_index, explorer_set=None):
        if model_index < 0:
            return self.default_policy
        by_broker = collections.OrderedDict()
        while self._pending:
            broker, explorer = self._pending.popleft()[:2]
            by_broker.setdefault(broker, set()).add(explorer)
        return [_restrict(_destination(b, list(ids)), explorer_set) for b, ids in by_broker.items()]
