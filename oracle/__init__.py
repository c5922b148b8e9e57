"""CPU oracle for the XingTian PPO/IMPALA learner-update hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything from this package, and there only as the *checker* -- never as the
thing that is measured or shipped.  The product path (``xingtian_amd``) never
imports ``oracle`` and fails loudly when its HIP extension is missing.

Parity status
-------------
* GAE (``oracle.returns.gae``): **pinned** -- checked bit-for-bit against the
  reference's own numpy implementation (``xt/agent/ppo/ppo.py:77-106``) executed
  under import stubs; the resulting vectors are committed in
  ``tests/golden/gae_*.npz`` together with ``oracle/gen_golden.py``.
* Host logic of the algorithm classes (``PPO``, ``IMPALAOpt``, the non-opt ``IMPALA`` with its numpy v-trace on
  probabilities) and the architecture tables: **pinned** -- ``oracle/gen_golden_alg.py`` / ``gen_golden_arch.py``
  EXECUTE the reference's own classes / functions under import stubs and commit what they produce
  (``tests/golden/alg_*.npz``, ``arch_tables.json``); the product's classes are tested against those fixtures.
* Everything the reference delegates to TensorFlow (conv/dense fwd+bwd, the PPO
  and v-trace losses, the Keras ``impala_loss``, clip_by_global_norm, AdamOptimizer / RMSProp / tf.keras Adam):
  **parity unpinned**.
  TensorFlow (1.15 / 2.3.1, un-vendored, not installable here) holds the
  arithmetic and the reference's tests pin no numbers for it, so this package
  restates TF's published semantics (VALID/SAME padding, NHWC/HWIO layouts,
  softmax-CE, TF1 Adam, clip_by_global_norm) in float64 numpy and is
  cross-checked against an independent torch-autograd float64 implementation
  (``oracle/torch_ref.py``) in ``tests/test_oracle.py``.
"""
