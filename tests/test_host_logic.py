"""CPU-only checks of the host side: the C-ABI library loads and exports every declared symbol, the model
mirrors the reference's parameter tree, flat-buffer packing, the dataset semantics, checkpoint discovery,
and that the product path refuses to run without the GPU (no compute calls are made here)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from dvae_amd import _lib
    names = ge.declared_symbols()
    assert len(names) >= 30
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    h = _lib.lib()
    for n in names:
        assert hasattr(h, n), n
    assert h.dvae_version() >= 100
    # pure-host helpers of the ABI (no device work)
    assert h.dvae_bn_ws_bytes(16384, 512, 2) > 0 and h.dvae_l1_ws_bytes(1000) > 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dvae_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        _lib.lib()


def test_model_mirrors_reference_parameter_tree(golden_dir):
    import dvae_amd
    g = np.load(os.path.join(golden_dir, "c0_b4_t64.npz"))
    m = dvae_amd.DisentangledVAE(speaker_size=4, latent_dim=32, batch_size=4)
    assert [n for n, _ in m.named_parameters()] == list(g["param_names"])
    assert sum(p.numel() for p in m.parameters()) == 61367680
    keys = set(m.state_dict().keys())
    for k in g.files:
        if k.startswith("bn_"):
            assert k[3:] in keys, k
    m128 = dvae_amd.DisentangledVAE(speaker_size=4, latent_dim=32, n_frames=128)
    assert sum(p.numel() for p in m128.parameters()) == 94930304
    assert len(m.backward_param_order()) == 84


def test_cpu_tensors_are_refused():
    import dvae_amd
    m = dvae_amd.DisentangledVAE(speaker_size=4, latent_dim=32)
    x = torch.zeros(2, 80, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, x)
    with pytest.raises(RuntimeError):
        m.decode(torch.zeros(2, 32))


def test_flat_adam_packing():
    from dvae_amd.optim import FlatAdam
    ps = [("a", torch.nn.Parameter(torch.randn(3, 5))), ("b", torch.nn.Parameter(torch.randn(7))),
          ("c", torch.nn.Parameter(torch.randn(2, 2, 2)))]
    before = [p.detach().clone() for _, p in ps]
    opt = FlatAdam(ps, lr=1e-3)
    assert opt.offsets == {"a": 0, "b": 16, "c": 24} and opt.numel == 32   # 16-byte aligned views
    assert opt.views_intact()
    for (_, p), b in zip(ps, before):
        assert torch.equal(p.detach(), b)
    ps[1][1].grad.add_(1.0)
    assert float(opt.flat_g[16:23].sum()) == 7.0 and float(opt.flat_g.sum()) == 7.0
    opt.zero_grad()
    assert float(opt.flat_g.abs().sum()) == 0.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        opt.step()
    sd = opt.state_dict()
    opt.load_state_dict(sd)
    assert opt.views_intact()


def test_dataset_pairs_same_speaker_and_crops(tmp_path):
    from dvae_amd.data import SpeechDatasetGVAE, write_synthetic_corpus
    root = write_synthetic_corpus(str(tmp_path / "corpus"), n_speakers=2, n_utt=8, length=96, seed=0)
    ds = SpeechDatasetGVAE(root, samples_length=64, seed=1)
    assert len(ds) == 8                                 # 2 speakers x (8 // 2) pairs  (dataset.py:63-76)
    for i in range(len(ds)):
        u1, u2 = ds.utterance_fp[i]
        assert os.path.dirname(u1) == os.path.dirname(u2) and u1 != u2
        m1, m2, spk = ds[i]
        assert m1.shape == (80, 64) and m2.shape == (80, 64) and m1.dtype == torch.float64
        assert int(spk) == ds.speaker_ids.index(os.path.basename(os.path.dirname(u1)))
    first = [tuple(p) for p in ds.utterance_fp]
    ds.shuffle_data()
    assert sorted(sum(([a, b] for a, b in first), [])) == sorted(sum(([a, b] for a, b in ds.utterance_fp), []))
    # shorter than the crop -> right zero padding (dataset.py:100-101); exactly the crop -> offset 0
    ds_pad = SpeechDatasetGVAE(root, samples_length=128, seed=1)
    m1, _, _ = ds_pad[0]
    assert m1.shape == (80, 128) and float(m1[:, 96:].abs().sum()) == 0.0
    ds_eq = SpeechDatasetGVAE(root, samples_length=96, seed=1)
    assert ds_eq[0][0].shape == (80, 96)


def test_dataset_against_reference_fixture(tmp_path, golden_dir):
    """data.SpeechDatasetGVAE vs the REAL preprocessing/dataset.py:53-114 class (tests/golden/make_golden.py dataset):
    same corpus, same np.random seed on the global generator -> same pair list after construction and after
    shuffle_data(), same crop offsets in item order (incl. the right zero-pad branch and an odd utterance count)."""
    from dvae_amd.data import SpeechDatasetGVAE, write_synthetic_corpus
    g = np.load(os.path.join(golden_dir, "dataset_pairs.npz"))
    T = int(g["samples_length"])
    root = write_synthetic_corpus(str(tmp_path / "corpus"), n_speakers=2, n_utt=8, length=96, seed=0)
    rs = np.random.RandomState(9)
    os.makedirs(os.path.join(root, "spk_short"))
    for u in range(5):
        np.save(os.path.join(root, "spk_short", f"utt{u:03d}_mel.npy"), rs.uniform(0, 1, size=(80, 40 + u)))
    np.random.seed(int(g["seed"]))
    ds = SpeechDatasetGVAE(root, samples_length=T)          # seed=None: draws from the global generator, as the reference
    assert ds.speaker_ids == list(g["speaker_ids"])

    def check(tag):
        pairs = g["pairs_" + tag]
        assert len(ds) == len(pairs)
        got = [[ds.speaker_ids.index(p.split("/")[-2]), int(os.path.basename(p)[3:6])] for row in ds.utterance_fp
               for p in row]
        assert np.array_equal(np.array(got).reshape(-1, 4), pairs)
        sums = []
        for i in range(len(ds)):
            m1, m2, spk = ds[i]
            assert int(spk) == int(g["labels_" + tag][i])
            for m, path, off in ((m1, ds.utterance_fp[i][0], g["offsets_" + tag][i][0]),
                                 (m2, ds.utterance_fp[i][1], g["offsets_" + tag][i][1])):
                full = np.load(path)
                want = np.pad(full, ((0, 0), (0, T - full.shape[1]))) if off < 0 else full[:, off:off + T]
                assert np.array_equal(m.numpy(), want), (tag, i, off)
                sums.append(float(m.sum()))
        np.testing.assert_allclose(np.array(sums), g["sums_" + tag], rtol=1e-12)

    check("epoch0")
    ds.shuffle_data()
    check("epoch1")


def test_init_matches_reference_init_weights():
    """init_weights (disentangled_vae.py:26-32, applied :195): every Linear xavier-uniform with bias 0.01, every Conv1d
    xavier-uniform with bias 0; LSTMs keep torch's default U(-1/sqrt(H), 1/sqrt(H)); BatchNorm gamma 1 / beta 0 /
    running stats (0, 1, 0)."""
    import math
    import dvae_amd
    torch.manual_seed(0)
    m = dvae_amd.DisentangledVAE(speaker_size=4, latent_dim=32, batch_size=4)
    n_lin = n_conv = n_lstm = n_bn = 0
    for name, p in m.named_parameters():
        if "lstm" in name:
            H = p.shape[0] // 4
            b = 1.0 / math.sqrt(H)
            p = p.detach()
            assert float(p.abs().max()) <= b * (1 + 1e-6) and float(p.abs().max()) > 0.9 * b, name
            if p.numel() >= 4096:                             # uniform, not e.g. normal clipped to the bound
                assert abs(float(p.mean())) < 0.05 * b and abs(float(p.std()) - b / math.sqrt(3)) < 0.05 * b, name
            n_lstm += 1
        elif name.endswith(".1.weight"):                      # BatchNorm gamma
            assert torch.equal(p.detach(), torch.ones_like(p)), name
            n_bn += 1
        elif name.endswith(".1.bias"):
            assert torch.equal(p.detach(), torch.zeros_like(p)), name
        elif p.dim() == 3:                                    # Conv1d weight, stored packed [5, Cout, Cin]
            assert p.shape[0] == 5 and tuple(m.state_dict()[name].shape) == (p.shape[1], p.shape[2], 5), name
            fan_in, fan_out = p.shape[2] * 5, p.shape[1] * 5
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            p = p.detach()
            assert float(p.abs().max()) <= bound * (1 + 1e-6) and float(p.abs().max()) > 0.95 * bound, name
            assert abs(float(p.std()) - bound / math.sqrt(3)) < 0.03 * bound, name
            n_conv += 1
        elif p.dim() == 2:                                    # Linear weight [out, in]
            bound = math.sqrt(6.0 / (p.shape[0] + p.shape[1]))
            p = p.detach()
            assert float(p.abs().max()) <= bound * (1 + 1e-6) and float(p.abs().max()) > 0.9 * bound, name
            assert abs(float(p.std()) - bound / math.sqrt(3)) < 0.05 * bound, name
            n_lin += 1
        elif "conv" in name or name.startswith("dec_modules"):   # Conv1d bias
            assert torch.equal(p.detach(), torch.zeros_like(p)), name
        else:                                                  # Linear bias
            assert torch.equal(p.detach(), torch.full_like(p, 0.01)), name
    assert (n_lin, n_conv, n_lstm, n_bn) == (6, 11, 28, 11)
    for name, b in m.named_buffers():
        want = 1.0 if name.endswith("running_var") else 0.0
        assert float(b.float().abs().max() if want == 0.0 else (b - 1).abs().max()) == 0.0, name


def test_load_last_model_picks_latest(tmp_path):
    import dvae_amd
    from dvae_amd.model.variational_base_vae import VariationalBaseModelVAE

    class Tiny(VariationalBaseModelVAE):
        def __init__(self):
            super().__init__("VCTK", 64, 80, 1, 32, 1e-4, torch.device("cpu"), 500, 4)
            self.model = torch.nn.Linear(2, 2)
            self.optimizer = None

    t = Tiny()
    assert t.load_last_model(str(tmp_path)) == 1
    for ep, val in ((5, 1.0), (500, 2.0), (20, 3.0)):
        sd = {k: torch.full_like(v, val) for k, v in t.model.state_dict().items()}
        torch.save(sd, tmp_path / f"DisentangledVAE_VCTK_{ep}.pth")
    assert t.load_last_model(str(tmp_path)) == 501      # variational_base_vae.py:144-149
    assert float(t.model.weight[0, 0]) == 2.0


def test_shard_range():
    from dvae_amd.ddp import shard_range
    assert [shard_range(512, r, 8) for r in (0, 7)] == [(0, 64), (448, 512)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_compute_mode_switch_is_host_side_state():
    """dvae_set_compute_mode / ops.set_compute_dtype: pure host state of the library, checkable without a GPU."""
    import dvae_amd  # noqa: F401
    from dvae_amd import ops
    from dvae_amd._lib import lib
    assert ops.get_compute_dtype() == ops.DEFAULT_COMPUTE_DTYPE == "fp32x3"
    try:
        ops.set_compute_dtype("bf16")
        assert lib().dvae_get_compute_mode() == 1 and ops.get_compute_dtype() == "bf16"
        with ops.compute_dtype("fp32"):
            assert ops.get_compute_dtype() == "fp32"
        assert ops.get_compute_dtype() == "bf16"
        assert lib().dvae_set_compute_mode(7) == -1          # DVAE_EINVAL, mode unchanged
        assert ops.get_compute_dtype() == "bf16"
        with pytest.raises(ValueError):
            ops.set_compute_dtype("fp8")
        ops.set_compute_dtype("fp32x3")
        assert lib().dvae_get_compute_mode() == 2
    finally:
        ops.set_compute_dtype(ops.DEFAULT_COMPUTE_DTYPE)


def test_split_k_chooser():
    """ops._split_k: whole rounds of the 512 resident workgroups, >= 256 of K per split, not only powers of two."""
    import dvae_amd  # noqa: F401
    from dvae_amd import ops
    assert ops._split_k(80, 16384) == 6            # 5 taps x 16 tiles of the 512->512 conv weight gradient
    assert ops._split_k(256, 16256) == 2           # W_hh gradient at H = 1024
    assert ops._split_k(4096, 1024) == 1           # already many rounds: no split
    for tiles in (1, 3, 16, 20, 80, 128, 500, 513, 5000):
        for k in (32, 255, 256, 1000, 16384, 100000):
            s = ops._split_k(tiles, k)
            assert s >= 1 and (s == 1 or k // s >= 256), (tiles, k, s)
    assert ops._split_k(513, 16384) > 1            # 513 tiles: a second, nearly empty round unless the work is split
    # the 256 x 128 kernels' model (conv weight gradients, Cout >= 256): one round of 256 slots, the atomic epilogue priced in
    assert ops._split_k(40, 16384, slots=256, fixed=384) == 6
    assert ops._split_k(40, 65536, slots=256, fixed=384) == 6      # (the 128-tile model said 19: 210 us against 162)
    # every split is one more atomic pass over the output: the tiny H = 64 weight gradients stop at k = 512 per split
    assert ops._split_k(2, 65280) == 128 and ops._split_k(2, 16256) == 63
    # the two 134 MB Linear weights keep their single, storing launch (FlatAdam.set_store_first depends on it)
    assert ops._split_k(ops._tiles(2048, 16384), 128) == 1 and ops._split_k(ops._tiles(16384, 2048), 128) == 1


def test_stacked_lstm_schedule_rules():
    import dvae_amd  # noqa: F401
    from dvae_amd.ops import LstmStack2Fn as S
    assert S.chunk(128) == 64 and S.chunk(64) == 32
    assert S.usable(128, 1024, 2, False) and S.usable(64, 512, 2, False)
    assert not S.usable(128, 1024, 1, False)       # one layer: nothing to stack
    assert not S.usable(128, 64, 2, True)          # the bidirectional H=64 encoder LSTM runs whole sequences per launch
    assert not S.usable(1, 1024, 2, False) and not S.usable(127, 1024, 2, False)


def test_frontend_refuses_cpu():
    import dvae_amd  # noqa: F401
    from dvae_amd.frontend import MelFrontend
    with pytest.raises(RuntimeError):
        MelFrontend(device="cpu")


def test_flat_adam_state_is_layout_independent():
    """The optimizer checkpoint holds the moments per parameter in the REFERENCE's layout: a conv weight stored packed
    [5][Cout][Cin] round-trips through [Cout][Cin][5].  A format-1 file (raw flat vectors) was written by TWO builds —
    conv slices in torch's layout, later packed — and says nowhere which: the loader refuses to guess and
    converts either variant when told (`legacy_layout`)."""
    import dvae_amd  # noqa: F401
    from dvae_amd.optim import FlatAdam

    class Layout:
        def reference_layout(self, name, t):
            return t.permute(1, 2, 0) if name == "conv.weight" else t

        def storage_layout(self, name, t):
            return t.permute(2, 0, 1) if name == "conv.weight" else t

    def make():
        conv = torch.nn.Parameter(torch.zeros(5, 3, 2))      # packed [K][Cout][Cin]
        lin = torch.nn.Parameter(torch.zeros(4, 6))
        return FlatAdam([("conv.weight", conv), ("lin.weight", lin)], lr=1e-3, layout=Layout())

    a = make()
    a.exp_avg.copy_(torch.arange(a.numel, dtype=torch.float32))
    a.exp_avg_sq.copy_(torch.arange(a.numel, dtype=torch.float32) * 2)
    a.dev_state[0] = 7
    sd = a.state_dict()
    assert sd["format"] == 2 and tuple(sd["exp_avg"]["conv.weight"].shape) == (3, 2, 5)
    assert torch.equal(sd["exp_avg"]["conv.weight"], a.exp_avg[:30].view(5, 3, 2).permute(1, 2, 0))
    b = make()
    b.load_state_dict(sd)
    for (_, va), (_, vb) in zip(a._moment_views(a.exp_avg), b._moment_views(b.exp_avg)):
        assert torch.equal(va, vb)
    for (_, va), (_, vb) in zip(a._moment_views(a.exp_avg_sq), b._moment_views(b.exp_avg_sq)):
        assert torch.equal(va, vb)
    assert b.t == 7
    # format 1: the conv slice of the flat vector is [Cout][Cin][5] (torch layout)
    ref = torch.arange(30, dtype=torch.float32).view(3, 2, 5)
    flat = torch.zeros(a.numel)
    flat[:30] = ref.reshape(-1)
    old = {"t": 3, "lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-8, "names": ["conv.weight", "lin.weight"],
           "exp_avg": flat, "exp_avg_sq": flat.clone()}
    c = make()
    with pytest.raises(ValueError, match="legacy_layout"):
        c.load_state_dict(old)                                  # nothing in the file says which variant it is
    c.load_state_dict(old, legacy_layout="torch")
    assert torch.equal(c.exp_avg[:30].view(5, 3, 2), ref.permute(2, 0, 1))
    # the variant the build right before format 2 wrote: the conv slice is already in STORAGE (packed) layout
    packed = torch.arange(30, dtype=torch.float32)
    flat_p = torch.zeros(a.numel)
    flat_p[:30] = packed
    d = make()
    d.load_state_dict(dict(old, exp_avg=flat_p, exp_avg_sq=flat_p.clone()), legacy_layout="packed")
    assert torch.equal(d.exp_avg[:30], packed)                  # unchanged: same storage layout then and now
    assert d.t == 3
    # a format-1 file for a model WITHOUT layout differences needs no hint
    e = FlatAdam([("lin.weight", torch.nn.Parameter(torch.zeros(4, 6)))], lr=1e-3)
    e.load_state_dict({"t": 1, "lr": 1e-3, "betas": (0.9, 0.999), "eps": 1e-8, "names": ["lin.weight"],
                       "exp_avg": torch.ones(24), "exp_avg_sq": torch.ones(24)})
    assert float(e.exp_avg.sum()) == 24.0
    with pytest.raises(ValueError):
        c.load_state_dict(dict(sd, format=3))


def test_attach_reducer_restores_the_zero_grad_folding_it_found():
    """A sharded reducer (rs_ag) switches the Adam launch's gradient clear off (a rank reads and could clear only its own
    slices); the next reducer that is not sharded — or None — brings back whatever the setting was BEFORE, including an explicit
    False of the caller (tests/_ddp2_gpu_child.py reads the summed gradients after the step)."""
    import dvae_amd  # noqa: F401
    from dvae_amd.model.variational_base_vae import VariationalBaseModelVAE
    from dvae_amd.optim import FlatAdam

    class Red:
        def __init__(self, mode):
            self.mode = mode

    class Tiny(VariationalBaseModelVAE):
        def __init__(self):
            super().__init__("VCTK", 64, 80, 1, 32, 1e-4, torch.device("cpu"), 500, 4)
            self.model = torch.nn.Linear(4, 4)
            self.optimizer = FlatAdam(list(self.model.named_parameters()), lr=1e-3)

    t = Tiny()
    opt = t.optimizer
    assert opt.fold_zero_grad is True
    t.attach_reducer(Red("all_reduce"))
    assert opt.fold_zero_grad is True
    t.attach_reducer(Red("rs_ag"))
    assert opt.fold_zero_grad is False
    t.attach_reducer(Red("rs_ag"))                       # twice in a row: the remembered value is still the original one
    t.attach_reducer(Red("all_reduce"))
    assert opt.fold_zero_grad is True
    t.attach_reducer(Red("rs_ag"))
    t.attach_reducer(None)
    assert opt.fold_zero_grad is True
    opt.fold_zero_grad = False                           # the caller's own choice ...
    t.attach_reducer(Red("all_reduce"))
    assert opt.fold_zero_grad is False                   # ... is not overridden by a reducer that is not sharded
    t.attach_reducer(Red("rs_ag"))
    t.attach_reducer(None)
    assert opt.fold_zero_grad is False                   # ... and comes back after a sharded one


def test_store_first_counters_restart_with_zero_grad():
    """a step that was abandoned after its backward pass (a graph capture that failed, an exception) must not
    leave the written-exactly-once count of a store-first gradient at 1 for the retry."""
    from dvae_amd.optim import FlatAdam
    ps = [("a", torch.nn.Parameter(torch.randn(8, 4))), ("b", torch.nn.Parameter(torch.randn(8)))]
    opt = FlatAdam(ps, lr=1e-3)
    opt.set_store_first(["a"])
    ps[0][1]._dvae_sf_writes = 1                         # an abandoned attempt's backward wrote it once
    opt.zero_grad()                                      # the retry starts a step
    assert ps[0][1]._dvae_sf_writes == 0
    assert not hasattr(ps[1][1], "_dvae_sf_writes") or ps[1][1]._dvae_sf_writes == 0
    ps[0][1]._dvae_sf_writes = 1                         # the retry's backward
    opt._store_first_guard()                             # exactly one write: fine, and the count is taken
    assert ps[0][1]._dvae_sf_writes == 0
    ps[0][1]._dvae_sf_writes = 2
    with pytest.raises(RuntimeError, match="store-first"):
        opt._store_first_guard()


def test_shared_gpu_decision_needs_local_world_size(monkeypatch):
    """WORLD_SIZE counts the ranks of every node; only LOCAL_WORLD_SIZE may switch the persistent recurrences off."""
    import dvae_amd  # noqa: F401
    from dvae_amd import ops
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setenv("WORLD_SIZE", "16")               # two nodes of eight: NOT a shared GPU
    assert ops._ranks_share_a_gpu() is False
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert ops._ranks_share_a_gpu() is False
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "16")
    assert ops._ranks_share_a_gpu() is True


def test_abort_process_group_never_blocks_or_raises():
    """train.main's failure path: tearing down the communicators of a failing rank must not itself enter a collective."""
    import dvae_amd  # noqa: F401
    from dvae_amd import train
    train._abort_process_group()                         # no process group: a no-op
    import inspect
    src = inspect.getsource(train.main)
    body = src[src.index("except BaseException"):]
    # (round 6: the abort runs on a thread of its own with a five-second bound, then os._exit(1) — ADVICE r5)
    assert "target=_abort_process_group" in body and "os._exit(1)" in body
    assert body.index("raise") < body.index("dist.barrier()")


def test_ctypes_structs_follow_the_header():
    """include/dvae_hip.h is the contract: the ctypes mirrors in _lib.py must list the same fields in the same order (a field
    inserted in one place only — round 6 added dvae_lstm_dir_t.dbias_part and dvae_slab_desc_t — shifts every pointer behind
    it), and the constants must agree."""
    import re
    import dvae_amd  # noqa: F401
    from dvae_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dvae_hip.h")).read()

    def fields(struct_name):
        body = hdr[:hdr.index("} " + struct_name + ";")]
        body = body[body.rindex("typedef struct {"):]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        body = body[body.index("{") + 1:]
        out = []
        for stmt in body.split(";"):                      # `int kind, d0, d1, d2` -> four fields
            for decl in stmt.split(","):
                m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(?:\[[^\]]*\])?\s*$", decl.strip())
                if m and decl.strip():
                    out.append(m.group(1))
        return out

    assert fields("dvae_lstm_dir_t") == [f[0] for f in _lib.LstmDir._fields_]
    assert fields("dvae_slab_desc_t") == [f[0] for f in _lib.SlabDesc._fields_]
    assert fields("dvae_repack_desc_t") == [f[0] for f in _lib.RepackDesc._fields_]
    assert int(re.search(r"#define DVAE_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION
    assert int(re.search(r"#define DVAE_SLAB_FOLD_MAX (\d+)", hdr).group(1)) == _lib.SLAB_FOLD_MAX
    assert int(re.search(r"#define DVAE_PERS_BIAS_SLABS (\d+)", hdr).group(1)) == _lib.PERS_BIAS_SLABS
    import ctypes as C
    assert C.sizeof(_lib.SlabDesc) == 40


def test_trajectory_band_is_monotone_and_floored(golden_dir):
    """conftest.trajectory_band: never below the 1e-4 of the single-step contract, non-decreasing over the steps inside each
    group of like losses, and the perturbed runs only widen it."""
    import numpy as np
    from conftest import trajectory_band, trajectory_band_bf16
    g = np.load(os.path.join(golden_dir, "trajectory_c0_b4_t64.npz"))
    ref, band = trajectory_band(g)
    _, thin = trajectory_band(g, perturbed=False)
    assert ref.shape == band.shape == (int(g["n_steps"]), 8)
    assert np.all(band >= 1e-4) and np.all(np.diff(band, axis=0) >= 0) and np.all(band >= thin)
    assert np.allclose(band[:, 0], band[:, 4]) and np.allclose(band[:, 5], band[:, 7])
    _, b16 = trajectory_band_bf16(g)
    assert np.all(b16 >= band) and np.all(b16 >= 2e-3)
    # what the fixture says about the reference itself: agrees with itself at step 1, is chaotic by step 5
    spread = np.abs(g["traj_fp32"] - g["traj_fp32"][-1][None]).max(0) / np.abs(g["traj_fp32"][-1])
    assert spread[0].max() < 1e-6 and spread[4].max() > 1e-2
