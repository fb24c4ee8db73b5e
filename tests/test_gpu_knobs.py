"""The library's A/B kernel-selection knobs (read once per process from the environment) each select a
different kernel family for the same call; every setting must pass the same operator parity cases.  One subprocess per
setting (the knobs are latched at first use), each running a representative subset of tests/test_gpu_ops.py: a pooled
encoder layer, a wide layer, the affine DUAL block in its plain and up-sampling form, the 3-channel ends."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SUBSET = "enc_conv2 or enc_conv5 or affine_blk1 or affine_blk7 or out_conv_f3 or enc_conv1_in3 or disc_conv1_k3"

KNOBS = [
    dict(CAPE_GEMM_BF16X6="0", CAPE_DW_BF16X6="0"),            # every contraction on the exact-fp32 MFMA kernels
    dict(CAPE_GEMM_BF16X6_DUAL="0"),                           # affine DUAL forward on the exact-fp32 kernel only
    dict(CAPE_DW_BF16X6="0"),                                  # weight gradient on the exact-fp32 kernels only
    dict(CAPE_GEMM_PLAIN="0", CAPE_DW_PLAIN="0"),              # generic gather kernels for every launch
    dict(CAPE_SPMM_UNROLL="0"),                                # sparse kernels: plain entry loop
    dict(CAPE_SPMM_UNROLL="8"),                                # sparse kernels: entries in unrolled groups of 8
]


@pytest.mark.parametrize("knobs", KNOBS, ids=lambda k: "+".join("%s=%s" % kv for kv in sorted(k.items())))
def test_operator_parity_under_knob(knobs):
    env = dict(os.environ)
    env.update(knobs)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-x", "-q", "-m", "gpu",
                        "-k", "test_cheb_conv_fwd_bwd and twopass and (%s)" % SUBSET],
                       env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
