"""Compile-time resource check of the hot kernels (tools/kernel_resources.py: hipcc -S for gfx950, no GPU needed).

Round 2 found 128 B/thread of scratch (private-memory) traffic in every bf16 conv launch -- 2-3x the bytes of C in rocprofv3's WRITE_SIZE --
because a statistics object was reached through a run-time pointer.  A non-zero ScratchSize in these files is a regression; so is a
plain 8-wave LDS-DMA GEMM that no longer fits two workgroups per CU (> 128 VGPRs)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('kernel_resources', os.path.join(ROOT, 'tools', 'kernel_resources.py'))
kr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(kr)

import sys  # noqa: E402
sys.path.insert(0, ROOT)
from carla_garage_amd._lib import SOURCES as HOT  # noqa: E402  every compiled source (a new file is covered the day it joins the build)


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
@pytest.mark.parametrize('fname', HOT)
def test_hot_kernels_use_no_scratch_and_keep_their_occupancy(fname):
  rows = kr.table(os.path.join(ROOT, 'carla_garage_amd', 'csrc', fname))
  assert rows
  names = kr.demangle([r[0] for r in rows])
  for name, vgpr, agpr, total, scratch, occupancy, lds in rows:
    assert scratch == 0, (names[name], scratch)
    if 'conv_gemm_glds_kernel<128, 128, 2, 2, 4, 2, false, true>' in names[name] or 'conv_gemm_glds_kernel<256, 128, 3, 4, 4, 2, false, true>' in names[name]:
      assert total <= 128, (names[name], total)  # two 8-wave workgroups / one 16-wave workgroup per CU


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='needs hipcc')
@pytest.mark.parametrize('fname', HOT)
def test_no_packed_fp32_valu_instructions(fname):
  """Round 3: v_pk_fma_f32 / v_pk_add_f32 (SLP-vectorised row sums of layernorm_bwd_kernel) returned wrong sums in waves that shared a CU with the
  MFMA weight-gradient kernel -- the 'result depends on the co-runner' issue of round 2.  The library is built without the vectorizers; no kernel
  may contain a packed FP32 VALU instruction."""
  bad = kr.packed_fp32_instructions(os.path.join(ROOT, 'carla_garage_amd', 'csrc', fname))
  assert not bad, {kr.demangle([k])[k]: v for k, v in bad.items()}
