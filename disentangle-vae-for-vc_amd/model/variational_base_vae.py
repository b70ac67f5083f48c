"""Training-loop surface of the reference's VariationalBaseModelVAE
(/root/reference/model/variational_base_vae.py:30-202), device-agnostic and data-parallel capable.

Kept: `step`, `train`, `run_training`, `load_last_model`, checkpoint naming
`DisentangledVAE_VCTK_{epoch}.pth` holding `model.state_dict()`.
Changed on purpose:
  * no hard-coded `.to("cuda")` (:81-82): batches go to `self.device`;
  * the 8 `.item()` host syncs of :70 become ONE device->host copy of the 8 scalars;
  * optional data parallelism: `attach_reducer()` installs ddp.GradReducer, after which `step`
    overlaps a bucketed RCCL all-reduce of the flat gradient buffer with backward;
  * TensorBoard is optional (tensorboardX is not a dependency): scalars go to `logs_path/scalars.jsonl`;
  * `estimate_trained_model` (:205-239) keeps its tensor part (eval-mode forward of one batch with train=False) and
    writes the mels as .npy (+ PNG when matplotlib is importable; librosa.display is not a dependency);
    `voice_conversion_mel` keeps its tensor part as `convert_mel`; vocoder / file I/O stay outside.
"""
from __future__ import annotations

import json
import os
from glob import glob
from pathlib import Path

import torch


def chunking_mel(melspectrogram, n_frames: int = 64, device="cuda"):
    """[80, L] (numpy or tensor) -> device tensor [L//T + 1, 80, T], tail zero-padded — the reference's chunking_mel
    (variational_base_vae.py:335-348), done by a HIP kernel."""
    from .._lib import check, lib, ptr, stream
    mel = torch.as_tensor(melspectrogram).to(device=device, dtype=torch.float32).contiguous()
    C, L = mel.shape
    n = L // n_frames + 1
    out = torch.empty((n, C, n_frames), device=mel.device, dtype=torch.float32)
    check(lib().dvae_mel_to_chunks(ptr(mel), ptr(out), C, L, n_frames, n, stream()), "dvae_mel_to_chunks")
    return out


class VariationalBaseModelVAE:
    def __init__(self, dataset, width, height, channels, latent_sz, learning_rate, device, log_interval, batch_size,
                 normalize=False, flatten=True):
        self.dataset = dataset
        self.width, self.height, self.channels = width, height, channels
        self.input_sz = (channels, width, height)
        self.latent_sz = latent_sz
        self.lr = learning_rate
        self.device = device
        self.log_interval = log_interval
        self.normalize_data = normalize
        self.flatten_data = flatten
        self.model = None       # set by subclasses
        self.optimizer = None
        self.batch_size = batch_size
        self.reducer = None     # ddp.GradReducer when data parallel
        # hipGraph replay of the whole train step: ~1 000 launches become one graph launch
        self._use_graph = False
        self._graph_ddp = True
        self._graph = None
        self._graph_sig = None
        self._graph_calls = 0

    def loss_function(self):
        raise NotImplementedError

    # ---- data parallel (new functionality: the reference is single-device)
    def attach_reducer(self, reducer):
        self.reducer = reducer
        if self.optimizer is not None:
            # a sharded step (rs_ag) reads — and could clear — only this rank's slices of the gradient buffer: zero_grad
            # launches every step while such a reducer is attached; whatever the setting was before comes back with the
            # next reducer that is not sharded (or with None), so switching modes back and forth leaves no trace
            opt = self.optimizer
            sharded = reducer is not None and getattr(reducer, "mode", "all_reduce") == "rs_ag"
            if sharded:
                if getattr(self, "_fold_before_shard", None) is None:
                    self._fold_before_shard = bool(opt.fold_zero_grad)
                if opt.fold_zero_grad:
                    opt.fold_zero_grad = False
                    opt._clean = False
            elif getattr(self, "_fold_before_shard", None) is not None:
                if opt.fold_zero_grad != self._fold_before_shard:
                    opt.fold_zero_grad = self._fold_before_shard
                    opt._clean = False
                self._fold_before_shard = None

    def enable_graph(self, flag: bool = True, ddp=None):
        """Capture the train step into a hipGraph on its second call and replay it afterwards.  The first call runs
        eagerly (it is a real step, warms everything up and, under data parallelism, lets RCCL set up its communicator
        outside the capture).  The graph bakes host scalars into kernel arguments (loss coefficients, 1/batch_size,
        BatchNorm train/eval, shapes): `_graph_signature` is compared on every step and the graph is re-captured when
        any of them changed, so `update_kl()` or `model.eval()` take effect exactly as in the eager path.  The learning
        rate is NOT baked in: it lives in the optimizer's device state (`FlatAdam.sync_scalars`, called before every
        replay), so `optimizer.param_groups[0]['lr'] = ...` every step — a schedule — costs one 8-byte copy when the
        value changed and never a re-capture.
        `ddp`: with a reducer attached the bucketed RCCL all-reduces can be captured INSIDE the graph (ranks replay the
        same step a single GPU does).  OPT-IN (ddp=True or DVAE_DDP_GRAPH=1): no run with two or more ranks has shown
        it equal to the eager data-parallel step yet; without it a data-parallel step runs eagerly whatever `flag` says.
        A capture that fails falls back to the eager step with a warning (`graph_fallback` holds the reason)."""
        self._use_graph = flag
        self._graph_ddp = (os.environ.get("DVAE_DDP_GRAPH", "0") == "1") if ddp is None else bool(ddp)
        self.graph_fallback = None
        self._graph = None
        self._graph_sig = None
        self._graph_calls = 0

    def _graph_signature(self, data1):
        from .. import ops
        opt = self.optimizer
        return (tuple(data1.shape), tuple(opt.betas), float(opt.eps),
                float(self.mse_cof), float(self.kl_cof), int(self.batch_size), bool(self.model.training),
                self.reducer is not None, getattr(self.reducer, "world_size", 1), ops.current_mode(),
                bool(ops.LSTM_PERSISTENT), bool(ops.deterministic()))

    def _eager_train_step(self, data1, data2):
        # (until round 5 the step ran inside an ops.ZeroArena: one clear launch for the outputs that split-k contractions
        # accumulated into atomically; split contractions store slabs now and nothing needs clearing)
        return self._train_step_body(data1, data2)

    def _train_step_body(self, data1, data2):
        self.optimizer.zero_grad()
        if hasattr(self, "losses_vector_full") and hasattr(self.model, "forward_full"):
            # the eight scalars as one vector (fused loss kernels) on the UNSPLIT outputs; backward is seeded with the
            # constant (1, 0, ..., 0): `vec[0].backward()` and the reference's ten output slices cost ~25 zero-fill /
            # copy launches of autograd glue per step
            vec = self.losses_vector_full(data1, data2, *self.model.forward_full(data1, data2))
            losses = None
        else:
            vec = None
            losses = self.loss_functionGVAE2(data1, data2, *self.model(data1, data2), train=True)
        if self.reducer is not None:
            self.reducer.begin()
        if vec is not None:
            if getattr(self, "_loss_seed", None) is None or self._loss_seed.device != vec.device:
                self._loss_seed = torch.zeros(8, device=vec.device)
                self._loss_seed[0] = 1.0
            torch.autograd.backward(vec, self._loss_seed)
        else:
            losses[0].backward()
        if self.reducer is not None:
            self.reducer.finish()
            self.reducer.step(self.optimizer)       # full Adam after an all-reduce, or the sharded step (mode "rs_ag")
        else:
            self.optimizer.step(grad_scale=1.0)
        return vec.detach() if vec is not None else torch.stack([l.detach() for l in losses])

    def _step_graph(self, data1, data2):
        m = self.model
        dev = data1.device
        Bh = data1.shape[0]
        S, Cn = m.speaker_size, m.latent_dim - m.speaker_size
        self._graph_calls += 1
        if self._graph_calls == 1:
            return self._eager_train_step(data1, data2)
        sig = self._graph_signature(data1)
        if self._graph is not None and sig != self._graph_sig:
            self._graph = None          # a baked-in scalar or the batch shape changed: capture again
        if self._graph is None:
            self._graph_sig = sig
            self._g_x1, self._g_x2 = torch.empty_like(data1), torch.empty_like(data2)
            # the two content-noise halves are views of ONE buffer: the model uses it as is (no concatenation launch)
            eps_c = torch.empty((2 * Bh, Cn), device=dev)
            self._g_eps_c = eps_c
            self._g_eps = (eps_c[:Bh], eps_c[Bh:], torch.empty((Bh, S), device=dev))
        self._g_x1.copy_(data1)
        self._g_x2.copy_(data2)
        user_eps = m.eps_override
        if user_eps is None:               # the eager step's two draws (content [2*Bh, Cn], style [Bh, S]), same order
            self._g_eps_c.normal_()
            self._g_eps[2].normal_()
        else:
            for dst, src in zip(self._g_eps, user_eps):
                dst.copy_(src.to(dev))
        m.eps_override = self._g_eps
        # learning rate / gradient scale: device scalars, refreshed OUTSIDE the captured region
        self.optimizer.sync_scalars(1.0 / self.reducer.world_size if self.reducer is not None else 1.0)
        if self._graph is not None and not getattr(self.optimizer, "_clean", True):
            # something accumulated into the gradient buffer since the last Adam launch (a manual backward between two
            # replays): a graph captured on a clean buffer holds no zero_grad launch of its own
            self.optimizer.zero_grad()
        try:
            if self._graph is None:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                # with a reducer the RCCL collectives are captured too (opt-in, see enable_graph); its watchdog thread
                # makes HIP calls of its own, hence thread-local capture checking
                # (also without a reducer while a process group exists — bench.py times the single-rank graph inside a
                # data-parallel run: RCCL's watchdog is there either way)
                import torch.distributed as _dist
                pg = _dist.is_available() and _dist.is_initialized()
                mode = {"capture_error_mode": "thread_local"} if (self.reducer is not None or pg) else {}
                if self.reducer is not None and pg:
                    # c10d's watchdog thread polls the end events of the EAGER collectives of the warm-up steps every 100 ms
                    # until it has seen them complete.  HIP refuses hipEventQuery on an event whose stream has meanwhile
                    # entered a capture (hipErrorCapturedEvent — RCCL's stream joins this capture), the watchdog rethrows
                    # and the process aborts: seen as a rare abort of the captured data-parallel step (2 of ~15 complete
                    # test-suite runs, round 6).  The device is idle here (synchronize above): give the watchdog a few
                    # periods to retire what it still holds.  (torch's own wait for pending event queries in
                    # CUDAGraph.capture_begin covers global-mode captures only.)
                    import time as _time
                    _time.sleep(0.5)
                try:
                    with torch.cuda.graph(g, **mode):
                        self._g_losses = self._eager_train_step(self._g_x1, self._g_x2)
                except Exception as e:      # nothing of the step has run: fall back to the eager step, for good
                    if self.reducer is None:
                        raise
                    import warnings
                    self.graph_fallback = repr(e)[:300]
                    warnings.warn("hipGraph capture of the data-parallel step failed; running eagerly: " + self.graph_fallback)
                    self._use_graph = False
                    self._graph = None
                    torch.cuda.synchronize()
                    return self._eager_train_step(self._g_x1, self._g_x2)
                self._graph = g
            self._graph.replay()
            if getattr(self.optimizer, "fold_zero_grad", False):
                self.optimizer._clean = True        # the replayed Adam launch cleared the ranges (host flags do not replay)
        finally:
            m.eps_override = user_eps
        return self._g_losses.clone()       # the static buffer is overwritten by the next replay

    def step_async(self, data1, data2, speaker_ids=None):
        """One TRAIN step without any host synchronisation: the 8 loss scalars come back as a device tensor, so the
        host can enqueue the next step (noise draw, input copies, graph launch) while this one runs.  `step(...,
        train=True)` is this plus one device->host copy."""
        if self._use_graph and (self.reducer is None or self._graph_ddp):
            return self._step_graph(data1, data2)
        return self._eager_train_step(data1, data2)

    # ---- variational_base_vae.py:58-70
    def step(self, data1, data2, speaker_ids, train=False):
        if train:
            out = tuple(self.step_async(data1, data2, speaker_ids).tolist())   # one D2H copy, not 8 .item() syncs
            self._check_and_recover()
            return out
        outs = self.model(data1, data2)
        losses = self.loss_functionGVAE2(data1, data2, *outs, train=train)
        return tuple(torch.stack([l.detach() for l in losses]).tolist())

    @staticmethod
    def _check_device_errors():
        """Raises if a bounded cross-workgroup wait of a persistent LSTM launch gave up since the last check (the host is
        synchronised at every call site of this)."""
        from .. import ops
        ops.lstm_pers_check()

    def _check_and_recover(self):
        try:
            self._check_device_errors()
        except Exception:
            # the Adam launches skipped their update (and their gradient clear) while the error word was set: the next
            # zero_grad must really clear
            if self.optimizer is not None and hasattr(self.optimizer, "_clean"):
                self.optimizer._clean = False
            raise

    # ---- variational_base_vae.py:74-101
    def train(self, train_loader, epoch, logging_func=print):
        self.model.train()
        # the running sums of :88-96 live on the device (fp64); the host reads them once per epoch instead of
        # stalling the queue after every step
        tot = torch.zeros(8, dtype=torch.float64, device=self.device)
        last = None
        for data1, data2, speaker_ids in train_loader:
            data1 = data1.to(self.device, non_blocking=True).float()
            data2 = data2.to(self.device, non_blocking=True).float()
            speaker_ids = speaker_ids.view(-1)
            last = self.step_async(data1, data2, speaker_ids)
            tot.add_(last)
        tot = tot.tolist()
        self._check_and_recover()
        last_style = float(last[7]) if last is not None else 0.0
        if hasattr(train_loader, "dataset") and hasattr(train_loader.dataset, "shuffle_data"):
            train_loader.dataset.shuffle_data()
        n_items = len(train_loader.dataset) if hasattr(train_loader, "dataset") else len(train_loader)
        logging_func("====> Epoch: {} Average loss: {:.4f}".format(epoch, tot[0] / max(1, n_items)))
        # NB the reference returns the style KL of the LAST batch, not the total (:101)
        return tot[1], tot[2], tot[3], tot[4], tot[5], tot[6], last_style

    # ---- variational_base_vae.py:127-149
    def load_last_model(self, checkpoints_path, logging_func=print):
        name = self.model.__class__.__name__
        ids = []
        for f in glob(f"{checkpoints_path}/*.pth"):
            parts = Path(f).stem.split("_")
            if len(parts) == 3 and parts[2].isdigit():
                ids.append((int(parts[2]), f))
        if not ids:
            logging_func(f"Training {name} model from scratch...")
            return 1
        start_epoch, last = max(ids, key=lambda it: it[0])
        self.model.load_state_dict(torch.load(last, map_location=self.device))
        opt = last[:-4] + ".opt"
        if os.path.exists(opt):
            sd = torch.load(opt, map_location="cpu")
            self.optimizer.load_state_dict(sd, legacy_layout=os.environ.get("DVAE_OPT_LEGACY_LAYOUT"))
            if sd.get("cuda_rng_state") is not None and torch.device(self.device).type == "cuda":
                g = torch.cuda.default_generators[torch.device(self.device).index or 0]
                if self.reducer is not None and self.reducer.rank > 0:
                    # the checkpoint holds rank 0's generator.  The other ranks keep THEIR seed (train.py / bench.py seed
                    # rank r with seed + 7919 r) and continue at the checkpointed Philox offset — every rank draws the
                    # same number of normals per step — instead of replaying rank 0's noise or their own from offset 0
                    seed = int(sd.get("cuda_rng_seed", g.initial_seed())) + 7919 * self.reducer.rank
                    g.manual_seed(seed)
                    if sd.get("cuda_rng_offset") is not None:
                        g.set_offset(int(sd["cuda_rng_offset"]))
                else:
                    torch.cuda.set_rng_state(sd["cuda_rng_state"], self.device)   # eps stream continues where it stopped
        logging_func(f"Loading {name} model from last checkpoint ({start_epoch})...")
        return start_epoch + 1

    def update_(self):
        pass

    # ---- variational_base_vae.py:205-239
    @torch.no_grad()
    def estimate_trained_model(self, test_loader, checkpoints_path, estimation_dir, n_show=5):
        """Load the last checkpoint, run ONE batch through the eval-mode forward (`train=False`: content z = mu,
        BatchNorm on running statistics) and store the first `n_show` original / reconstructed mels of x1 as
        `{epoch}_original_mel_{i}.npy` / `{epoch}_recons_mel_{i}.npy` (the reference stores specshow PNGs; PNGs are
        also written here when matplotlib imports).  Returns (epoch, recons_x1, recons_x2)."""
        logging_epoch = self.load_last_model(checkpoints_path, logging_func=lambda *_: None)
        was_training = self.model.training
        self.model.eval()
        os.makedirs(estimation_dir, exist_ok=True)
        try:
            # a loader with sample_batch (data.GpuPairLoader) hands out a batch WITHOUT advancing its epoch streams: under
            # data parallelism only rank 0 comes here and must not fall out of step with the other ranks' permutations
            data1, data2, _ = test_loader.sample_batch() if hasattr(test_loader, "sample_batch") else next(iter(test_loader))
            data1 = data1.to(self.device).float()
            data2 = data2.to(self.device).float()
            outs = self.model(data1, data2, train=False)
            recons_x1, recons_x2 = outs[2], outs[3]
            try:
                import matplotlib
                matplotlib.use("Agg")
                import matplotlib.pyplot as plt
            except Exception:                      # plotting is optional
                plt = None
            for i in range(min(n_show, data1.shape[0])):
                for tag, mel in (("original", data1[i]), ("recons", recons_x1[i])):
                    arr = mel.detach().cpu().numpy()
                    base = os.path.join(estimation_dir, f"{logging_epoch}_{tag}_mel_{i}")
                    import numpy as np
                    np.save(base + ".npy", arr)
                    if plt is not None:
                        plt.figure()
                        plt.title(("reconstructed" if tag == "recons" else "original") + " mel spectrogram")
                        plt.imshow(arr, origin="lower", aspect="auto")
                        plt.colorbar(format="%f")
                        plt.savefig(base + ".png")
                        plt.close()
            return logging_epoch, recons_x1, recons_x2
        finally:
            self.model.train(was_training)

    # ---- tensor part of voice_conversion_mel (variational_base_vae.py:269-301); file I/O, plots and the WaveNet
    # vocoder of the reference stay outside
    @torch.no_grad()
    def convert_mel(self, source_mel, target_mel):
        """source_mel, target_mel: [80, L] -> dict(source, recons, converted, spectral_detail), each [80, n*T].
        Content of the source, style (mean style_mu over chunks) of the target; eval-mode BatchNorm."""
        from .._lib import check, lib, ptr, stream
        m = self.model
        was_training = m.training
        m.eval()
        try:
            T, S, Cn = m.n_frames, m.speaker_size, m.latent_dim - m.speaker_size
            src = chunking_mel(source_mel, T, self.device)
            trg = chunking_mel(target_mel, T, self.device)
            n, k = src.shape[0], trg.shape[0]
            src_style, src_content = m.encode_heads(src)      # [n, 2S], [n, 2Cn] (mu | logvar)
            trg_style, _ = m.encode_heads(trg)
            z_src = torch.empty((n, S + Cn), device=src.device, dtype=torch.float32)
            z_conv = torch.empty_like(z_src)
            L = lib()
            check(L.dvae_conversion_latents(ptr(src_style), ptr(src_content), ptr(trg_style), ptr(z_src), ptr(z_conv),
                                            n, k, S, Cn, stream()), "dvae_conversion_latents")
            recons = m.decode(z_src)
            conv = m.postnet.forward_plus_input(m.decode(z_conv))

            def cat(x, clamp):
                out = torch.empty((x.shape[1], n * T), device=x.device, dtype=torch.float32)
                check(L.dvae_chunks_to_mel(ptr(x.contiguous()), ptr(out), n, x.shape[1], T, 0.0, 1.0, int(clamp),
                                           stream()), "dvae_chunks_to_mel")
                return out
            source, recons_v, conv_v = cat(src, False), cat(recons, False), cat(conv, True)
            detail = torch.empty_like(source)
            check(L.dvae_mul_div(ptr(source), ptr(recons_v), ptr(conv_v), ptr(detail), source.numel(), stream()),
                  "dvae_mul_div")
            return {"source": source, "recons": recons_v, "converted": conv_v, "spectral_detail": detail}
        finally:
            m.train(was_training)

    def _is_rank0(self):
        return self.reducer is None or self.reducer.rank == 0

    # ---- variational_base_vae.py:156-202
    def run_training(self, train_loader, test_loader, epochs, report_interval, sample_sz=64, reload_model=True,
                     checkpoints_path="", logs_path="", images_path="", estimation_dir="", logging_func=print,
                     start_epoch=None):
        start_epoch = self.load_last_model(checkpoints_path, logging_func) if reload_model else 1
        run_name = "DisentangledVAE_VCTK"
        log_f = None
        if logs_path and self._is_rank0():
            os.makedirs(os.path.join(logs_path, run_name), exist_ok=True)
            log_f = open(os.path.join(logs_path, run_name, "scalars.jsonl"), "a")
        history = []
        for epoch in range(start_epoch, start_epoch + epochs):
            r1, r2, r1h, r2h, k1, k2, ks = self.train(train_loader, epoch, logging_func)
            nb = max(1, len(train_loader))
            rec = {"epoch": epoch, "Loss/Reconstruction Loss1": r1 / nb, "Loss/Reconstruction Loss2": r2 / nb,
                   "Loss/Reconstruction Loss1 hat": r1h / nb, "Loss/Reconstruction Loss2 hat": r2h / nb,
                   "Loss/Z1 KL Loss": k1 / nb, "Loss/Z2 KL Loss": k2 / nb, "Loss/Z KL Style": ks / nb}
            history.append(rec)
            if self._is_rank0():
                logging_func(json.dumps(rec))
                if log_f:
                    log_f.write(json.dumps(rec) + "\n")
                    log_f.flush()
            if epoch % report_interval == 0 and checkpoints_path and self.reducer is not None \
                    and hasattr(self.reducer, "gather_moments"):
                self.reducer.gather_moments(self.optimizer)      # a collective: every rank, before rank 0 writes
            if epoch % report_interval == 0 and self._is_rank0() and checkpoints_path:
                os.makedirs(checkpoints_path, exist_ok=True)
                with torch.no_grad():
                    base = f"{checkpoints_path}/{run_name}_{epoch}"
                    torch.save(self.model.state_dict(), base + ".pth")      # reference format: weights only
                    osd = self.optimizer.state_dict()                        # added: Adam moments + step count
                    osd["cuda_rng_state"] = torch.cuda.get_rng_state(self.device)   # + the eps generator state
                    gen = torch.cuda.default_generators[torch.device(self.device).index or 0]
                    osd["cuda_rng_seed"], osd["cuda_rng_offset"] = int(gen.initial_seed()), int(gen.get_offset())
                    torch.save(osd, base + ".opt")
                # variational_base_vae.py:196-201: reconstructions of one test batch from the checkpoint just written
                if estimation_dir and test_loader is not None:
                    self.estimate_trained_model(test_loader, checkpoints_path, estimation_dir)
        if log_f:
            log_f.close()
        return history
