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
* The TensorFlow-side FORMULAS the reference itself writes down -- the PPO actor/critic losses
  (``xt/model/ppo/__init__.py:4-25``), ``CategoricalDist`` / ``DiagGaussianDist`` (``xt/model/tf_dist.py:49-130``),
  v-trace (``xt/model/impala/vtrace.py:39-115``), ``split_batches`` + ``vtrace_loss`` and its parts
  (``xt/model/impala/impala_cnn_opt.py:171-196, 299-351``), the Keras-form ``impala_loss`` closures
  (``xt/model/impala/impala_cnn.py:99-108``, ``impala_mlp.py:84-93``; ``K`` = a three-function torch-float64
  backend stand-in, ``tests/golden/tf_keras_impala_*.npz``): **pinned** -- ``oracle/gen_golden_tf.py`` EXECUTES those
  sources unmodified under a torch-float64 ``tf`` stand-in (``oracle/tf_shim.py``: only primitive ops are restated,
  each citing the TF 1.15 op / gradient function it follows) and commits values + autograd gradients as
  ``tests/golden/tf_*.npz`` (ratio exactly at 1 +- eps, |v - old_v| = VF_CLIP, done at t = 0 / T-2 / everywhere,
  A in {2, 4, 6, 18}, the BASELINE shapes (T=128, B=1/4) and (T=50, B=20, A=6)); ``tests/test_oracle.py`` holds the
  numpy restatement to ~1e-12 of them, ``tests/test_gpu_kernels.py`` the HIP loss kernels to fp32 tolerance.
* What TensorFlow's own LIBRARY computes (Conv2D / Dense forward and backward, what Keras' ``model.compile`` does around the ``impala_loss``
  closure (batch mean, 'mse', loss_weights), ``clip_by_global_norm``, ``AdamOptimizer`` / ``RMSPropOptimizer`` / ``tf.keras`` Adam
  update rules): **parity unpinned** -- TensorFlow (1.15 / 2.3.1, un-vendored, not installable here) holds that
  arithmetic and the reference's tests pin no numbers for it, so this package restates TF's published semantics
  (VALID / asymmetric-SAME padding, NHWC / HWIO layouts, TF1 Adam with eps outside the bias correction,
  ``g * clip / max(norm, clip)``) in float64 numpy with hand-derived gradients and is cross-checked against an
  independent torch-autograd float64 implementation (``oracle/torch_ref.py``) in ``tests/test_oracle.py``.
"""
