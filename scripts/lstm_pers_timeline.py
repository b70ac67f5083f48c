"""GPU box: in-kernel timeline of the persistent LSTM launches (dev build with -DDVAE_PERS_TS, DVAE_LIB_PATH).
usage: lstm_pers_timeline.py H N [bid] [fwd|bwd]  — median ns between stamps per wave over the frames of one workgroup:
0 frame start, 1 poll matched + barrier A, 2 MFMAs done / partial tiles written, 3 after barrier B, 4 epilogue done,
5 after barrier C, 6 (wave 0) payload drained."""
import os as _os
# needs the DEVELOPMENT build of the library (csrc/build.sh dev): probes / environment knobs / timelines are not in the product
_os.environ.setdefault("DVAE_LIB_PATH", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                                                       "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dvae_amd  # noqa: F401
from dvae_amd import _lib, ops
from dvae_amd._lib import check, lib, ptr, stream
from dvae_amd.derived import lstm_local

H, N = int(sys.argv[1]), int(sys.argv[2])
bid = int(sys.argv[3]) if len(sys.argv) > 3 else 5
bwd = len(sys.argv) > 4 and sys.argv[4] == "bwd"
T = 128
MODE = int(os.environ.get("LSTM_MODE", "1"))      # 1 bf16 (bf16 state), 2 fp32x3 (forward only, fp32 state)
L = lib()
L.dvae_lstm_pers_set_ts.argtypes = [C.c_void_p, C.c_int]
f = dict(device="cuda", dtype=torch.float32)
w_hh = torch.randn(4 * H, H, **f) / H ** 0.5
der = lstm_local(torch.zeros(4 * H, 64, **f), w_hh, torch.zeros(4 * H, **f), torch.zeros(4 * H, **f), MODE)
gates0 = torch.randn(T * N, 4 * H, **f) * 0.5
sdt = torch.bfloat16 if MODE == 1 else torch.float32
gates, h, c = torch.empty_like(gates0), torch.empty(T * N, H, device="cuda", dtype=sdt), torch.empty(T * N, H, **f)
dh, dg = torch.randn(T * N, H, **f) * 0.1, torch.empty(T * N, 4 * H, device="cuda", dtype=sdt)
dc = torch.empty(N, H, **f)
ts = torch.zeros(T * 8 * 8, device="cuda", dtype=torch.int64)
ws = ops.lstm_pers_workspace("cuda")


def dirs(b):
    d = (_lib.LstmDir * 1)()
    d[0].gates, d[0].c_all, d[0].h_out = ptr(gates), ptr(c), ptr(h)
    d[0].w_hh, d[0].w_packed = (ptr(der.w_hh_t), ptr(der.pack_b)) if b else (ptr(w_hh), ptr(der.pack_f))
    d[0].dh_out, d[0].dgates, d[0].dc_ws = ptr(dh), ptr(dg), ptr(dc)
    d[0].packed_mode, d[0].state_bf16, d[0].pers_ws = MODE, int(MODE == 1), ptr(ws)
    return d


for it in range(3):
    gates.copy_(gates0)
    L.dvae_lstm_pers_set_ts(ptr(ts) if (it == 2 and not bwd) else None, bid)
    check(L.dvae_lstm_seq_fwd(dirs(False), 1, T, N, H, H, stream()), "fwd")
    L.dvae_lstm_pers_set_ts(ptr(ts) if (it == 2 and bwd) else None, bid)
    check(L.dvae_lstm_seq_bwd(dirs(True), 1, T, N, H, H, stream()), "bwd")
torch.cuda.synchronize()
ops.lstm_pers_check()
s = ts.cpu().reshape(T, 8, 8).double() * 10.0      # ns (100 MHz)
nw = 8 if float(s[5, 7, 0]) > 0 else 4
pw = nw - 1
print(f"{'bwd' if bwd else 'fwd'} H={H} N={N} workgroup {bid} ({nw} waves): frame period median "
      f"{float((s[3:, 0, 0] - s[2:-1, 0, 0]).median()):.0f} ns")
names = ["poll+barA (0->1)", "loads+MFMA+red (1->2)", "barB (2->3)", "epilogue (3->4)", "barC (4->5)"]
for w in (0, 1, pw):
    row = [float((s[2:-1, w, p + 1] - s[2:-1, w, p]).median()) for p in range(5)]
    txt = ", ".join(f"{n} {v:.0f}" for n, v in zip(names, row))
    if w == 0:
        txt += f", payload+drain (5->6) {float((s[2:-1, 0, 6] - s[2:-1, 0, 5]).median()):.0f}"
        txt += f", own flag -> next barA {float((s[3:, 0, 1] - s[2:-1, 0, 6]).median()):.0f}"
    print(f"wave {w}: " + txt)
if os.environ.get("ALL_STAMPS"):
    # k-split backward (lstm_pers_bwd_x3k): 0 frame start, 1 dG flags seen + barrier A, 2 MFMAs done + cross-wave tiles written,
    # 3 own n-tile summed + partial tile published, 4 partial flags seen + barrier D, 5 barrier E, 6 epilogue + barrier F,
    # 7 (wave 0) dG payload drained
    for w in range(4):
        row = []
        for p in range(7):
            a, b = s[2:-1, w, p], s[2:-1, w, p + 1]
            ok = (a > 0) & (b > 0)
            row.append(float((b - a)[ok].median()) if ok.any() else float("nan"))
        print(f"wave {w}: " + ", ".join(f"{p}->{p + 1} {v:.0f}" for p, v in enumerate(row)))
    print(f"wave 0: own dG flag (7) -> next barrier A (1): {float((s[3:, 0, 1] - s[2:-1, 0, 7]).median()):.0f}")
