"""GPU-box: register-only fp32 MFMA chains -> the matrix-pipe ceiling of this chip at the clock it holds."""
import os as _os
# needs the DEVELOPMENT build of the library (csrc/build.sh dev): probes / environment knobs / timelines are not in the product
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa
from dvae_amd._lib import lib, check, stream
L = lib()
out = torch.zeros(4, device="cuda")
for shape, flop in ((32, 32 * 32 * 2 * 2), (16, 16 * 16 * 4 * 2)):
    for blocks in (256, 512, 1024):
        iters = 20000 if shape == 32 else 40000
        check(L.dvae_probe_mfma(blocks, 100, shape, out.data_ptr(), stream()), "probe"); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(L.dvae_probe_mfma(blocks, iters, shape, out.data_ptr(), stream()), "probe"); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = blocks * 4 * iters * 4 * flop / ms / 1e9
        print(f"shape {shape}: {blocks} blocks x 4 waves, {iters} iters: {ms:.2f} ms -> {tf:.1f} TFLOP/s")

# bf16 pipe with the split-mode stream (2x2 tiles x 6 partial products of v_mfma_f32_32x32x16_bf16 per k-step)
o2 = torch.zeros(2, device="cuda", dtype=torch.int64)
for pattern, nm in ((0, "zero operands"), (1, "random, constant"), (2, "random, changing every k-step")):
    for blocks in (256, 512):
        iters = 20000
        check(L.dvae_probe_mfma_bf16(blocks, 200, pattern, out.data_ptr(), o2.data_ptr(), stream()), "probe"); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); check(L.dvae_probe_mfma_bf16(blocks, iters, pattern, out.data_ptr(), o2.data_ptr(), stream()), "probe"); e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = blocks * 4 * iters * 24 * (32 * 32 * 16 * 2) / ms / 1e9
        cyc, ref = [int(v) for v in o2.tolist()]
        print(f"bf16 32x32x16, {nm}: {blocks} blocks x 4 waves: {ms:.2f} ms -> {tf:.0f} TFLOP/s bf16 ({tf / 6:.1f} split-mode fp32 "
              f"equivalent); s_memtime/s_memrealtime = {cyc / max(1, ref):.2f} (x 100 MHz)")
