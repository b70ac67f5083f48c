"""`python -m dvae_amd.selftest [--rounds 200] [--device 0]` — does THIS GPU run the W_hh-resident persistent recurrences right?

The persistent LSTM launches (csrc/lstm_pers.hip) hand h[t] / dG[t] from workgroup to workgroup through write-through stores,
flags and L1-bypassing loads: a form that is measured valid on gfx950, not architecturally guaranteed, and in
rounds 4 and 5 one chip in fifty-nine showed wrong rows in a kernel form that every other chip ran 5 000 times without one.
Before a long training run on a new box, this runs every persistent kernel the product dispatches (default arithmetic: forward +
both backward forms at H = 512 and H = 1024; bf16 mode: H = 512 and H = 1024, at T = 48 and at the T = 512 of BASELINE configs[4]) for `--rounds` rounds while a second stream
streams 0 .. 4 GiB through HBM, and compares every output word with the one-launch-per-frame kernels of the same arithmetic and,
bit for bit, with the first persistent round.  Exit code 0: all equal; 1: a mismatch (the report names kernel, tensor, round,
frame and rows, and the GPU's KFD unique_id); a failing box should train with DVAE_LSTM_PERSISTENT=0.
"""
from __future__ import annotations

import argparse
import glob
import sys

import os
import torch


def gpu_unique_ids():
    """KFD unique_id of every GPU node the topology shows (no HIP call)."""
    out = []
    for path in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
        try:
            kv = dict(ln.split()[:2] for ln in open(path) if len(ln.split()) >= 2)
            if int(kv.get("simd_count", "0")) > 0:
                out.append(hex(int(kv.get("unique_id", "0"))))
        except Exception:
            pass
    return out


def _pass(mode, H, T, N, pers, seed=0):
    """forward + backward recurrence of one layer through the C ABI; returns (gates, c, h, dgates)"""
    from . import _lib, ops
    from .derived import lstm_local
    L, st, ptr = _lib.lib(), _lib.stream(), _lib.ptr
    bf = mode == _lib.MODE_BF16
    g = torch.Generator(device="cuda").manual_seed(1000 + seed)
    f = dict(device="cuda", dtype=torch.float32)
    sdt = torch.bfloat16 if bf else torch.float32
    w_hh = (torch.rand(4 * H, H, generator=g, **f) * 2 - 1) / H ** 0.5
    der = lstm_local(torch.zeros(4 * H, 64, **f), w_hh, torch.zeros(4 * H, **f), torch.zeros(4 * H, **f), mode)
    gates = torch.rand(T * N, 4 * H, generator=g, **f) * 2 - 1
    dh = (torch.rand(T * N, H, generator=g, **f) * 2 - 1) * 0.1
    h = torch.full((T * N, H), float("nan"), device="cuda", dtype=sdt)
    c = torch.empty(T * N, H, **f)
    dg = torch.full((T * N, 4 * H), float("nan"), device="cuda", dtype=sdt)
    dc, db = torch.empty(N, H, **f), torch.zeros(2, 4 * H, **f)
    ws = ops.lstm_pers_workspace("cuda")
    d = (_lib.LstmDir * 1)()
    d[0].gates, d[0].c_all, d[0].h_out, d[0].w_hh, d[0].w_packed = ptr(gates), ptr(c), ptr(h), ptr(w_hh), ptr(der.pack_f)
    d[0].packed_mode, d[0].state_bf16 = mode, int(bf)
    b = (_lib.LstmDir * 1)()
    b[0].gates, b[0].c_all, b[0].w_hh, b[0].w_packed = ptr(gates), ptr(c), ptr(der.w_hh_t), ptr(der.pack_b)
    b[0].dh_out, b[0].dgates, b[0].dc_ws, b[0].packed_mode, b[0].state_bf16 = ptr(dh), ptr(dg), ptr(dc), mode, int(bf)
    if pers:
        ops._pers_claim("cuda")          # one persistent launch at a time per device, as every product call site
        d[0].pers_ws = b[0].pers_ws = ptr(ws)
        b[0].dbias_ih, b[0].dbias_hh = ptr(db[0]), ptr(db[1])
    _lib.check(L.dvae_lstm_seq_fwd(d, 1, T, N, H, H, st), "fwd")
    _lib.check(L.dvae_lstm_seq_bwd(b, 1, T, N, H, H, st), "bwd")
    if pers:
        ops.lstm_pers_check()
    return [t.float() for t in (gates, c, h, dg)]


def run(rounds: int = 200, log=print) -> int:
    from . import _lib, ops
    if not torch.cuda.is_available():
        raise RuntimeError("dvae_amd.selftest needs the GPU it is to test")
    log(f"GPU unique_id(s): {gpu_unique_ids()}; device {torch.cuda.current_device()}: {torch.cuda.get_device_name()}")
    cases = [("fp32x3", _lib.MODE_F32X3, 1024, 64, 128, 2e-5), ("fp32x3", _lib.MODE_F32X3, 512, 64, 128, 2e-5),
             ("bf16", _lib.MODE_BF16, 1024, 48, 256, 2e-2), ("bf16", _lib.MODE_BF16, 1024, 48, 128, 2e-2),
             ("bf16", _lib.MODE_BF16, 512, 48, 128, 2e-2),
             # BASELINE configs[4]'s per-GPU shape: the 16-row bf16 forms over T = 512 frames
             ("bf16", _lib.MODE_BF16, 1024, 512, 128, 2e-2), ("bf16", _lib.MODE_BF16, 512, 512, 128, 2e-2)]
    side = torch.cuda.Stream()
    a = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
    bbuf = torch.empty_like(a)
    bad, ran, skipped = 0, 0, 0
    names = ("gates", "c", "h", "dgates")
    prev = ops.get_compute_dtype()
    try:
        for name, mode, H, T, N, rtol in cases:
            ops.set_compute_dtype(name)
            if not (ops.lstm_persistent_usable(N, H, mode) and ops.lstm_persistent_usable(N, H, mode, bwd=True)):
                log(f"{name} H={H} N={N}: no persistent kernel on this device (skipped)")
                skipped += 1
                continue
            ran += 1
            ref = [t.clone() for t in _pass(mode, H, T, N, False)]
            first, case_bad = None, 0
            n_rounds = rounds if T <= 128 else max(1, rounds // 4)      # the long sequences cost 4-8 x per round
            for rnd in range(n_rounds):
                with torch.cuda.stream(side):
                    for _ in range(rnd % 5):
                        bbuf.copy_(a)
                try:
                    got = _pass(mode, H, T, N, True)
                except _lib.DvaeHipError as e:           # a bounded wait gave up: a bad round, not a traceback
                    log(f"MISMATCH {name} H={H} N={N}: round {rnd}: {e}")
                    case_bad += 1
                    continue
                clean = True
                for tn, x, y in zip(names, got, ref):
                    err = (x - y).abs()
                    tol = rtol * float(y.abs().max())
                    wrong = (not bool(torch.isfinite(x).all())) or float(err.max()) > tol
                    if not wrong and first is not None:
                        wrong = not torch.equal(x, first[names.index(tn)])
                    if wrong:
                        e = err.reshape(T, N, -1)
                        frames = (e.amax(dim=(1, 2)) > tol).nonzero().flatten().tolist()
                        rows = (e.amax(dim=(0, 2)) > tol).nonzero().flatten().tolist()
                        log(f"MISMATCH {name} H={H} N={N}: {tn}, round {rnd} ({rnd % 5} GiB of foreign traffic): max |diff| "
                            f"{float(err.max()):.3e} (tolerance {tol:.1e}); frames {frames[:6]}.. ({len(frames)}), rows {rows[:8]}.. ({len(rows)})")
                        case_bad += 1
                        clean = False
                        break
                if first is None and clean:              # the bitwise reference is a round that matched the frame kernels
                    first = [t.clone() for t in got]
            torch.cuda.synchronize()
            log(f"{name} H={H} N={N} T={T}: {case_bad} bad rounds of {n_rounds}")
            bad += case_bad
    finally:
        ops.set_compute_dtype(prev)      # leave the compute mode as it was found, whatever happened
    try:
        log(f"XCD-local hand-offs (row groups on one XCD, verified per launch): {ops.lstm_pers_local_launches()} launches so far"
            + (" — switched off (DVAE_PERS_XCD_LOCAL=0)" if os.environ.get("DVAE_PERS_XCD_LOCAL", "1")[:1] == "0" else ""))
    except Exception as e:      # statistics only
        log(f"XCD-local statistics unavailable: {e}")
    log(f"selftest cases: {ran} run, {skipped} skipped")
    log("selftest " + ("PASSED" if bad == 0 else f"FAILED: {bad} bad rounds — train on this GPU with DVAE_LSTM_PERSISTENT=0 and report its unique_id"))
    return 0 if bad == 0 else 1


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--rounds", type=int, default=200)
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    torch.cuda.set_device(args.device)
    return run(args.rounds)


if __name__ == "__main__":
    sys.exit(main())
