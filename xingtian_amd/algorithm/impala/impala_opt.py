"""``IMPALAOpt``: the v-trace learner of the reference's IMPALA configurations (xt/algorithm/impala/impala_opt.py:37-147):
rollout messages are concatenated in arrival order and fed to ``Model.train`` in sequential BATCH_SIZE chunks (no
shuffling: chunk boundaries must fall on whole trajectories); the reported loss is the mean over the chunks."""
import os

import numpy as np

from xingtian_amd.algorithm.algorithm import Algorithm, RolloutFields
from xingtian_amd.algorithm.alg_utils import FIFODistPolicy
from xingtian_amd.algorithm.impala.default_config import BATCH_SIZE
from xingtian_amd.register import Registers, import_config


@Registers.algorithm
class IMPALAOpt(Algorithm):
    FIELDS = ("cur_state", "logit", "action", "done", "reward")

    def __init__(self, model_info, alg_config, **kwargs):
        import_config(globals(), alg_config)
        actor_info = dict(model_info["actor"])
        actor_info.setdefault("max_batch", BATCH_SIZE)      # the HIP model sizes its workspace for one chunk
        super().__init__(alg_name="impala", model_info=actor_info, alg_config=alg_config)
        self.async_flag = False
        self._rollout = RolloutFields(*self.FIELDS)
        self._streamed = 0
        # asynchronous actors: new weights go back to whoever delivered the data of this update
        self.dist_model_policy = FIFODistPolicy(alg_config["instance_num"], prepare_times=self._prepare_times_per_train)

    # list-per-field names of the reference (impala_opt.py:43-47)
    states = property(lambda self: self._rollout.parts["cur_state"])
    behavior_logits = property(lambda self: self._rollout.parts["logit"])
    actions = property(lambda self: self._rollout.parts["action"])
    dones = property(lambda self: self._rollout.parts["done"])
    rewards = property(lambda self: self._rollout.parts["reward"])

    @staticmethod
    def _data_proc(episode_data):
        """The agent ships done / reward as python lists: they become bool / float64 arrays here."""
        return (episode_data["cur_state"], episode_data["logit"], episode_data["action"],
                np.asarray(episode_data["done"], dtype=bool), np.asarray(episode_data["reward"]))

    # ---- asynchronous ingest (transport.Prefetcher): the H2D of train k+1's messages under the GPU's train k
    def stage_message(self, train_data, ctr_info=None):
        """Called on the Prefetcher's thread AS A MESSAGE ARRIVES: decode + pinned staging (or DMA straight out of a pinned
        transport slot) + asynchronous H2D start now; ``prepare_data`` of the learner thread later only books the message
        (``{"_prefetched": rows}``).  Same data, same order, same train -- only the copy happens earlier.  -> rows staged."""
        if self.dp is not None and self.dp.feed == "round_robin":
            raise RuntimeError("IMPALAOpt.stage_message: DP_FEED round_robin filters messages in prepare_data; hand every "
                               "rank its own source (DP_FEED sharded) to prefetch")
        if not (getattr(self.actor, "stream_ingest", False) and hasattr(self.actor, "ingest_message")):
            raise RuntimeError("IMPALAOpt.stage_message needs the streaming ingest (model_config STREAM_INGEST)")
        fields = self._data_proc(train_data)
        ctr = ctr_info or {}
        self.actor.ingest_message(*fields, pinned=bool(ctr.get("_pinned_views")), slot_guard=ctr.get("_slot_guard"))
        return int(np.asarray(fields[0]).shape[0])

    def stage_group_complete(self):
        """(staging thread) the last message of a train has been staged: its label block follows the frames to HBM and the
        buffer set's copies-done event is recorded now, not between ``train()`` and the GPU's first kernel"""
        self.actor._ingest_obj().seal()

    def staged_generation(self):
        """trains whose buffer set the learner thread has taken over (the Prefetcher stays at most one train ahead)"""
        return self.actor.ingest_generation()

    def stage_inline_capable(self):
        """does the model call an inline prefetcher's hook while the device trains?  (the deferred in-graph tail of
        ``ImpalaCnnOpt.train_ingested``: synchronous loss, IO_TAIL_IN_GRAPH on, hipGraph or not)"""
        a = self.actor
        return bool(getattr(a, "stream_ingest", False) and getattr(a, "io_tail_in_graph", 0) and not getattr(a, "async_loss", False)
                    and hasattr(getattr(a, "net", None), "impala_wait_loss"))

    def stage_inline(self, hook):
        """an inline ``transport.Prefetcher``: ``hook()`` stages one waiting message (-> did it?) and is called by the model
        between two looks at the loss while the device trains; None detaches"""
        self.actor._ingest_obj().on_finish = None
        self.actor.net.idle_gate = None
        self.actor.net.idle_hook = hook

    def stage_thread_init(self, wake=None, idle=None, bind_device=True):
        """first call on the staging thread: bind it to the learner's device; ``wake`` is called whenever the learner takes
        over a buffer set; ``idle`` (threading.Event) is SET by the model while the learner thread waits for the GPU -- the
        staging thread does its Python work then instead of fighting the learner for the GIL"""
        if bind_device:
            import torch
            torch.cuda.set_device(self.actor.net.device)
        self.actor._ingest_obj().on_finish = wake
        self.actor.net.idle_gate = idle

    def prepare_data(self, train_data, **kwargs):
        if "_prefetched" in train_data:
            # staged by a transport.Prefetcher when it arrived: the frames are (on their way) in HBM, only the books remain
            self._streamed += 1
            self._rollout.add(**{k: None for k in self.FIELDS})
            return
        if self.dp is not None and not self.dp.takes(train_data):
            return                  # DP_FEED round_robin: this message belongs to another learner rank
        fields = self._data_proc(train_data)
        if getattr(self.actor, "stream_ingest", False) and hasattr(self.actor, "ingest_message"):
            ctr = kwargs.get("ctr_info") or {}
            pinned = bool(ctr.get("_pinned_views"))
            self.actor.ingest_message(*fields, pinned=pinned, slot_guard=ctr.get("_slot_guard"))   # async H2D starts now
            self._streamed += 1
            self._rollout.add(**{k: None for k in self.FIELDS})   # no reference kept: the arrays may be transport views
            return
        self._rollout.add(**dict(zip(self.FIELDS, fields)))

    def train(self, **kwargs):
        if self._streamed > 0 and self._streamed == len(self._rollout):
            # the whole rollout already sits in HBM: every BATCH_SIZE chunk in one C call, no concat, no upload
            loss = self.actor.train_ingested(BATCH_SIZE)
        else:
            if self._streamed:
                raise RuntimeError("IMPALAOpt.train: the rollout was only partly streamed to the device")
            states, *labels = self._rollout.stacked()
            losses = []
            for lo in range(0, len(states), BATCH_SIZE):
                hi = lo + BATCH_SIZE
                losses.append(self.actor.train(states[lo:hi], [x[lo:hi] for x in labels]))
            loss = np.mean(losses)
        self._rollout.reset()
        self._streamed = 0
        if self.dp is not None:
            self.dp.new_rollout()
        return loss

    def predict(self, state):
        return self.actor.predict(state)

    def save(self, model_path, model_index):
        """File NAME only, and no underscore in the stem, as the reference's IMPALAOpt (impala_opt.py:108-114)."""
        saved = self.actor.save_model(os.path.join(model_path, "actor" + str(model_index).zfill(5)))
        return [saved.split("/")[-1]]
