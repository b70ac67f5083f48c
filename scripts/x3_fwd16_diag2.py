"""GPU box, DEV library: what the consumers of the 16-row fp32x3 forward recurrence (H = 512, DVAE_PERS_X3_MT1=3) HAD IN THEIR
REGISTERS when a round under foreign HBM traffic comes back wrong.  Every workgroup dumps, per frame and thread, an xor-fold of
the h fragments it loaded, the pre-activations it used and the gate sums it formed; one workgroup per row group dumps its raw
fragments.  A quiet run gives the reference dumps (persistent runs are bitwise repeatable); a bad round is compared against it
dump by dump, earliest frame first, so the FIRST thing that differs is named: bytes that arrived wrong (and what they were:
the poison the ring was filled with = read before arrival; the previous occupant of the slot = stale line; something else),
pre-activations read wrong, or right inputs and wrong sums.
usage: x3_fwd16_diag2.py [--rounds 200] [--T 96] [--N 128] [--nslot 2|T] [--uncached] [--poison] [--nodump] [--nobwd] [--mt1 3]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=200)
ap.add_argument("--T", type=int, default=96)
ap.add_argument("--N", type=int, default=128)
ap.add_argument("--H", type=int, default=512)
ap.add_argument("--nslot", type=int, default=2)
ap.add_argument("--uncached", action="store_true", help="ring + flags in fine-grained memory (hipExtMallocWithFlags)")
ap.add_argument("--poison", action="store_true", help="fill the ring with a NaN pattern in front of every round")
ap.add_argument("--nodump", action="store_true")
ap.add_argument("--nobwd", action="store_true")
ap.add_argument("--mt1", type=int, default=3)
ap.add_argument("--maxbad", type=int, default=4)
ap.add_argument("--sent", type=int, default=0, help="1: the sentinel hand-off kernel (DVAE_PERS_SENT; --nslot > 4: a slot per frame)")
ap.add_argument("--traffic", default="big", help="big: rnd %% 5 copies of 1 GiB in front of the round (x3_handoff_stress.py); "
                "small: 48 copies of 32 MiB (a kernel boundary on the other stream every ~15 us of the launch); "
                "delay: big + the round's own launches held back by rnd*37 %% 400 us of filler so that the copies' boundaries "
                "fall inside the forward launch; none")
ap.add_argument("--fresh", action="store_true", help="allocate gates / h / c / dgates anew every round, as x3_handoff_stress.py's _x3_pass does "
                "(the one form in which the sentinel reproducer has failed so far)")
ap.add_argument("--realloc", action="store_true", help="draw the gates anew every round (torch.rand, as x3_handoff_stress.py does)")
args = ap.parse_args()
os.environ.setdefault("DVAE_LIB_PATH", os.path.join(ROOT, "disentangle-vae-for-vc_amd", "libdvae_dev.so"))
os.environ["DVAE_PERS_X3_MT1"] = str(args.mt1)
os.environ["DVAE_PERS_SENT"] = str(args.sent)
sys.path.insert(0, ROOT)
import torch
import dvae_amd  # noqa: F401
from dvae_amd import _lib, ops
from dvae_amd.derived import lstm_local

H, T, N = args.H, args.T, args.N
L, st, ptr = _lib.lib(), _lib.stream(), _lib.ptr
X3 = _lib.MODE_F32X3
XCH_OFF = 64 * 1024 + 4096
MT = 1 if (args.mt1 & 1) else 2
n_rb = (N + 16 * MT - 1) // (16 * MT)
grid = (H // 16) * n_rb
NU = (H // 32 // 4) * MT            # (chunk, row tile) units per wave
POISON = 0x7FA57FA5

g = torch.Generator(device="cuda").manual_seed(1000)
f = dict(device="cuda", dtype=torch.float32)
w_hh = (torch.rand(4 * H, H, generator=g, **f) * 2 - 1) / H ** 0.5
der = lstm_local(torch.zeros(4 * H, 64, **f), w_hh, torch.zeros(4 * H, **f), torch.zeros(4 * H, **f), X3)
gates0 = torch.rand(T * N, 4 * H, generator=g, **f) * 2 - 1
dh = (torch.rand(T * N, H, generator=g, **f) * 2 - 1) * 0.1
gates = torch.empty_like(gates0)
h, c = torch.empty(T * N, H, **f), torch.empty(T * N, H, **f)
dg, dc, db = torch.empty(T * N, 4 * H, **f), torch.empty(N, H, **f), torch.zeros(2, 4 * H, **f)

ws_bytes = int(L.dvae_lstm_pers_ws_bytes_slots(N, H, max(args.nslot, 4)))
hip = None
if args.uncached:
    # the HIP runtime torch already loaded (a second copy of the runtime in one process would be another runtime)
    path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)
    hip = C.CDLL(path)
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    p = C.c_void_p()
    rc = hip.hipExtMallocWithFlags(C.byref(p), C.c_size_t(ws_bytes), C.c_uint(0x1))      # hipDeviceMallocFinegrained
    assert rc == 0, rc
    hip.hipMemset(p, 0, C.c_size_t(ws_bytes))
    ws_ptr = p.value
    ws_t = None
else:
    ws_t = torch.zeros(ws_bytes, device="cuda", dtype=torch.uint8)
    ws_ptr = ws_t.data_ptr()
def gpu_ids():
    """unique_id of the GPU nodes of this box (KFD topology): which chip a failure was seen on"""
    import glob
    out = []
    for path in sorted(glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")):
        try:
            kv = dict(l.split()[:2] for l in open(path) if len(l.split()) >= 2)
            if int(kv.get("simd_count", "0")) > 0:
                out.append(hex(int(kv.get("unique_id", "0"))))
        except Exception:
            pass
    return out


print(f"box: {os.uname().nodename} gpu unique_id {gpu_ids()}", flush=True)
print(f"H={H} T={T} N={N} MT={MT} grid={grid} nslot={args.nslot} ws={ws_bytes >> 10} KiB uncached={args.uncached} poison={args.poison}",
      flush=True)

dbg = dbg_frag = None
if not args.nodump:
    dbg = torch.zeros(T * grid * 256 * 12, device="cuda", dtype=torch.int32)
    dbg_frag = torch.zeros(T * n_rb * 4 * NU * 3 * 64 * 4, device="cuda", dtype=torch.int32)
DBG_JB = 5


def run(pers, dump):
    global gates, h, c, dg
    if args.fresh and pers:
        junk = [torch.empty(1 << 20, device="cuda") for _ in range(3)]      # perturb the allocator's reuse pattern a little
        gates, h, c = torch.empty_like(gates0), torch.empty(T * N, H, **f), torch.empty(T * N, H, **f)
        dg = torch.empty(T * N, 4 * H, **f)
        del junk
    if args.realloc and pers:
        g2 = torch.Generator(device="cuda").manual_seed(1000)
        _ = (torch.rand(4 * H, H, generator=g2, **f) * 2 - 1) / H ** 0.5
        gates.copy_(torch.rand(T * N, 4 * H, generator=g2, **f) * 2 - 1)
    else:
        gates.copy_(gates0)
    h.fill_(float("nan"))
    d = (_lib.LstmDir * 1)()
    d[0].gates, d[0].c_all, d[0].h_out, d[0].w_hh, d[0].w_packed = ptr(gates), ptr(c), ptr(h), ptr(w_hh), ptr(der.pack_f)
    d[0].packed_mode = X3
    b = (_lib.LstmDir * 1)()
    b[0].gates, b[0].c_all, b[0].w_hh, b[0].w_packed = ptr(gates), ptr(c), ptr(der.w_hh_t), ptr(der.pack_b)
    b[0].dh_out, b[0].dgates, b[0].dc_ws, b[0].packed_mode = ptr(dh), ptr(dg), ptr(dc), X3
    if pers:
        d[0].pers_ws = b[0].pers_ws = ws_ptr
        b[0].dbias_ih, b[0].dbias_hh = ptr(db[0]), ptr(db[1])
        if args.poison and ws_t is not None:
            ws_t[XCH_OFF:].view(torch.int32).fill_(POISON)
        L.dvae_lstm_pers_set_dbg(ptr(dbg) if dump else None, ptr(dbg_frag) if dump else None, DBG_JB, args.nslot)
    _lib.check(L.dvae_lstm_seq_fwd(d, 1, T, N, H, H, st), "fwd")
    if not args.nobwd:
        _lib.check(L.dvae_lstm_seq_bwd(b, 1, T, N, H, H, st), "bwd")
    if pers:
        info = (C.c_int * 4)()
        rc = L.dvae_lstm_pers_check(ws_ptr, info, st)
        assert rc == 0, ("gave up", list(info))
    return gates, h


ref_g, ref_h = (t.clone() for t in run(False, False))
tol = 2e-5 * float(ref_g.abs().max())
dump = not args.nodump
for attempt in range(5):
    q_g, q_h = run(True, dump)
    torch.cuda.synchronize()
    if float((q_g - ref_g).abs().max()) <= tol:
        break
    print("the QUIET persistent run is wrong too; again", flush=True)
else:
    sys.exit("no clean quiet run")
if dump:
    ref_dbg = dbg.clone().reshape(T, grid, 256, 12)
    ref_frag = dbg_frag.clone().reshape(T, n_rb, 4, NU * 3, 64, 4)
q_g = q_g.clone()


def lanes_of(mask256):
    t = mask256.nonzero().flatten().tolist()
    return f"{len(t)} threads " + (f"[{t[0]}..{t[-1]}] lanes&63 {sorted(set(x & 63 for x in t))[:20]}" if t else "")


def analyse(rnd):
    d = dbg.reshape(T, grid, 256, 12)
    diff = d != ref_dbg                                   # [T, grid, 256, 12]
    per_frame = diff.reshape(T, -1).any(dim=1).nonzero().flatten().tolist()
    if not per_frame:
        print("     no per-thread dump differs from the quiet run" + (" (the sentinel kernel writes none)" if args.sent else
              ": the inputs AND the sums were right, the outputs went wrong after that"))
    else:
        print(f"     first frame with a differing dump: {per_frame[0]} (frames that differ: {len(per_frame)})")
    for ff in per_frame[:2]:
        for name, sl in (("pre-activations used (x)", slice(4, 8)), ("fragment xor-fold (h loaded)", slice(0, 4)), ("gate sums", slice(8, 12))):
            dd = diff[ff, :, :, sl].any(dim=2)           # [grid, 256]
            wgs = dd.any(dim=1).nonzero().flatten().tolist()
            if not wgs:
                print(f"       frame {ff}: {name}: equal")
                continue
            rbs = sorted(set(w % n_rb for w in wgs))
            print(f"       frame {ff}: {name}: {len(wgs)} workgroups differ (row groups {rbs}; jb {sorted(set(w // n_rb for w in wgs))[:40]}); "
                  f"first {wgs[0]}: {lanes_of(dd[wgs[0]])}")
    # the raw fragments of the dumping workgroup of the first bad row group
    fr = dbg_frag.reshape(T, n_rb, 4, NU * 3, 64, 4)
    fd = fr != ref_frag
    ff_list = fd.reshape(T, -1).any(dim=1).nonzero().flatten().tolist()
    if not ff_list:
        print("       raw fragments of the dumping workgroups: all equal to the quiet run")
        return
    ff = ff_list[0]
    idx = fd[ff].nonzero()
    print(f"       raw fragments first differ at frame {ff}: {idx.shape[0]} words; (rb, wave, unit*3+plane, lane, word) of the first 12:")
    for row in idx[:12].tolist():
        rb_, w_, up_, ln_, wd_ = row
        got, want = int(fr[ff, rb_, w_, up_, ln_, wd_]) & 0xFFFFFFFF, int(ref_frag[ff, rb_, w_, up_, ln_, wd_]) & 0xFFFFFFFF
        note = " = POISON" if got == POISON else ""
        # was it the word this position held in an earlier frame (a stale line)?
        for back in range(1, 7):
            if ff - back >= 1 and (int(ref_frag[ff - back, rb_, w_, up_, ln_, wd_]) & 0xFFFFFFFF) == got:
                note += f" = this position's word of frame {ff - back} (-{back})"
                break
        print(f"         rb {rb_} wave {w_} frag {up_} lane {ln_} (row {ln_ & 15}, q {ln_ >> 4}) word {wd_}: got {got:08x} want {want:08x}{note}")
    bad_lanes = sorted(set(idx[:, 3].tolist()))
    bad_frags = sorted(set(idx[:, 2].tolist()))
    bad_waves = sorted(set(idx[:, 1].tolist()))
    print(f"       bad lanes {bad_lanes}; fragments {bad_frags}; waves {bad_waves}; row groups {sorted(set(idx[:, 0].tolist()))}")


side = torch.cuda.Stream()
a = torch.empty(1 << 28, device="cuda", dtype=torch.float32)
b = torch.empty_like(a)
filler = torch.zeros(1 << 22, device="cuda")
bad = 0
for rnd in range(args.rounds):
    with torch.cuda.stream(side):
        if args.traffic in ("big", "delay"):
            for _ in range(rnd % 5):
                b.copy_(a)
        elif args.traffic == "small":
            for i in range(48):
                b[i << 23:(i + 1) << 23].copy_(a[i << 23:(i + 1) << 23])
    if args.traffic == "delay":
        for _ in range((rnd * 37) % 400 // 8):
            filler.add_(1.0)                      # ~8 us each on the launch stream: shifts the launches against the copies
    got_g, got_h = run(True, dump)
    e = (got_g - ref_g).abs().reshape(T, N, -1)
    if float(e.max()) <= tol:
        if not torch.equal(got_g, q_g):
            print(f"round {rnd}: within tolerance but NOT bitwise equal to the quiet run", flush=True)
        continue
    bad += 1
    frames = (e.amax(dim=(1, 2)) > tol).nonzero().flatten().tolist()
    f0 = frames[0]
    rows = (e[f0].amax(dim=1) > tol).nonzero().flatten().tolist()
    cols = (e[f0].amax(dim=0) > tol).nonzero().flatten().tolist()
    print(f"round {rnd} ({rnd % 5} GiB): OUTPUT first bad frame {f0} of {len(frames)}, bad rows {rows}, bad cols {len(cols)} "
          f"[{cols[0]}..{cols[-1]}], max err {float(e[f0].max()):.3e}", flush=True)
    if dump:
        torch.cuda.synchronize()
        analyse(rnd)
    if bad >= args.maxbad:
        break
torch.cuda.synchronize()
print(f"RESULT sent={args.sent} traffic={args.traffic} realloc={args.realloc} nslot={args.nslot} uncached={args.uncached} poison={args.poison} dump={dump} bwd={not args.nobwd}: {bad} bad rounds of {rnd + 1}")
