"""Weight-distribution policies with the reference's behaviour (xt/algorithm/alg_utils.py:26-117)."""
from collections import deque, defaultdict


def _clip_explorer_id(raw_dist_info, clip_set):
    if not clip_set:
        return raw_dist_info
    elif raw_dist_info["explorer_id"] == -1:
        raw_dist_info["explorer_id"] = clip_set
    else:
        raw_dist_info["explorer_id"] = [_id for _id in raw_dist_info["explorer_id"] if _id in clip_set]
    return raw_dist_info


class DefaultAlgDistPolicy(object):
    def __init__(self, actor_num, **kwargs):
        self.actor_num = actor_num
        self.default_policy = {"broker_id": -1, "explorer_id": -1}

    def get_dist_info(self, model_index, explorer_set=None):
        return _clip_explorer_id(self.default_policy, explorer_set)

    def add_processed_ctr_info(self, ctr_info):
        pass


class DivideDistPolicy(DefaultAlgDistPolicy):
    def get_dist_info(self, model_index, explorer_set=None):
        if model_index > -1:
            self.default_policy.update({"explorer_id": model_index % self.actor_num})
        return _clip_explorer_id(self.default_policy, explorer_set)


def _fetch_broker_info(ctr_relation_buf):
    ctr_list = list()
    default_policy = {"broker_id": -1, "explorer_id": -1}
    for _broker, _explorer in ctr_relation_buf.items():
        default_policy.update({"broker_id": _broker, "explorer_id": list(_explorer)})
        ctr_list.append(default_policy.copy())
    return ctr_list


class FIFODistPolicy(DefaultAlgDistPolicy):
    """Distribute to whoever submitted explore data."""

    def __init__(self, actor_num, prepare_times, **kwargs):
        super(FIFODistPolicy, self).__init__(actor_num, **kwargs)
        self._processed_agent = deque()
        self.prepare_data_times = prepare_times

    def add_processed_ctr_info(self, ctr_info):
        self._processed_agent.append(ctr_info)

    def get_dist_info(self, model_index, explorer_set=None):
        if model_index < 0:
            return self.default_policy
        ctr_relation_buf = defaultdict(set)
        for _ in range(len(self._processed_agent)):
            _info = self._processed_agent.popleft()
            ctr_relation_buf[_info[0]].update((_info[1],))
        infos = _fetch_broker_info(ctr_relation_buf)
        return [_clip_explorer_id(i, explorer_set) for i in infos] if explorer_set else infos
