"""CPU oracle for the adaptive PnP-ADMM SCI hot path.  *** TEST INFRASTRUCTURE ONLY ***

This package is a plain NumPy / PyTorch-CPU restatement of the reference algorithm
(xyvirtualgroup/AdaptivePnP_SCI, dvp_linear_inv_2_stage_ADMM_tensor_online.py and its callees);
every function cites the reference file:line it follows.  It exists so that the HIP path can be
checked, and so that bench.py can time a CPU baseline next to the GPU number.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import it.  The
product package `adaptivepnp_sci_amd` never imports it and has no CPU fallback: without the
HIP library it raises.

Parity pin: PINNED.  Every function here is checked bit-for-bit (rel-L2 == 0.0) in the build
container against the imported reference itself (tools/make_golden.py drives /root/reference
through tools/ref_shim.py) and against the golden vectors committed under tests/golden/ that
the same script wrote; tests/test_oracle_golden.py re-checks the committed vectors on CPU.
The one third-party routine on the path, scikit-image's `denoise_tv_chambolle` (pinned 0.18.1 by
the reference's readme.md:15, not vendored), is restated from its published algorithm in
oracle/tv_chambolle.py and pinned against goldens generated from scikit-image 0.18.3's own
source (the reference holds no test vectors of its own for this routine).
"""
