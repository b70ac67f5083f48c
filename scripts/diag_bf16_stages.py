"""GPU-box: stage-by-stage forward agreement, HIP bf16 mode vs oracle/bf16_ref, each stage fed the ORACLE's input."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import dvae_amd
from dvae_amd import ops
from dvae_amd.ops import ConvBnActFn, LinearFn, ACT_RELU, ACT_NONE, ACT_TANH
from oracle import bf16_ref as R
from oracle.fill import fill_state_dict, synthetic_pair
B, T = 2, 64
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
if mode == "fp32":
    R.r16 = lambda t: t
x1, _ = synthetic_pair(B, T, 21)
m = R.RefDVAEBf16(4, 32, T); m.load_state_dict(fill_state_dict(m.state_dict())); m.train()
ops.set_compute_dtype(mode)
w = dvae_amd.ConvolutionalMulVAE("VCTK", T, 80, 32, 1e-4, 0.01, 500, False, batch_size=B, speaker_size=4,
                                 device=torch.device("cuda"), latent_dim=32, mse_cof=10, kl_cof=10)
w.model.load_state_dict(fill_state_dict(w.model.state_dict())); w.model.train()
hm = w.model
def frames(x):   # [B,C,T] -> [T*B, C]
    return x.permute(2, 0, 1).reshape(T * x.shape[0], x.shape[1]).contiguous().cuda()
def unframes(y, C):
    return y.cpu().reshape(T, B, C).permute(1, 2, 0)
def d(a, b):
    return f"{float((a - b).norm()) / float(b.norm()):.2e} (max {float((a - b).abs().max()):.1e}, scale {float(b.abs().max()):.1e})"
with torch.no_grad():
    x = x1
    for i, (blk, hblk) in enumerate(zip(m.enc_modules, hm.enc_modules)):
        ref_pre = R.conv5(x, R._child(blk[0]).weight, R._child(blk[0]).bias)
        ref = F.relu(blk[1](ref_pre))
        c, bn = hblk[0].conv if hasattr(hblk[0], "conv") else hblk[0], hblk[1]
        got = ConvBnActFn.apply(frames(x), c.weight, c.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked, None, B, 1, ACT_RELU, True)
        print(f"enc conv block {i}: {d(unframes(got, 512), ref)}")
        x = ref
    seq = R.lstm(m.enc_lstm, x.transpose(1, 2))                 # [B,T,128]
    got = hm._lstm(hm.enc_lstm, frames(x), T, B)                # [T*B,128]
    print("enc_lstm:", d(unframes(got, 128), seq.transpose(1, 2)))
    flat = seq.reshape(B, -1)
    ref = F.relu(R._lin(m.enc_linear, flat))
    lin = hm.enc_linear.linear_layer
    got = LinearFn.apply(flat.cuda().contiguous(), lin.weight, lin.bias, ACT_RELU)
    print("enc_linear:", d(got.cpu(), ref))
    z = torch.randn(B, 32, generator=torch.Generator().manual_seed(3))
    h1 = R._lin(m.dec_pre_linear1, z); h2 = R._lin(m.dec_pre_linear2, h1)
    g1 = LinearFn.apply(z.cuda(), hm.dec_pre_linear1.weight, hm.dec_pre_linear1.bias, ACT_NONE)
    print("dec_pre_linear1:", d(g1.cpu(), h1))
    g2 = LinearFn.apply(h1.cuda(), hm.dec_pre_linear2.weight, hm.dec_pre_linear2.bias, ACT_NONE)
    print("dec_pre_linear2:", d(g2.cpu(), h2))
    hh = h2.view(B, T, 128)
    ref = R.lstm(m.dec_lstm1, hh)                                # [B,T,512]
    got = hm._lstm(hm.dec_lstm1, frames(hh.transpose(1, 2)), T, B)
    print("dec_lstm1:", d(unframes(got, 512), ref.transpose(1, 2)))
    x = ref.transpose(1, 2)
    for i, (blk, hblk) in enumerate(zip(m.dec_modules, hm.dec_modules)):
        ref = F.relu(R._conv_bn(blk, x))
        c, bn = hblk[0], hblk[1]
        got = ConvBnActFn.apply(frames(x), c.weight, c.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                bn.num_batches_tracked, None, B, 1, ACT_RELU, True)
        print(f"dec conv block {i}: {d(unframes(got, 512), ref)}")
        x = ref
    ref = R.lstm(m.dec_lstm2, x.transpose(1, 2))
    got = hm._lstm(hm.dec_lstm2, frames(x), T, B)
    print("dec_lstm2:", d(unframes(got, 1024), ref.transpose(1, 2)))
    y = R._lin(m.dec_linear2, ref)
    lin = hm.dec_linear2.linear_layer
    got = LinearFn.apply(frames(ref.transpose(1, 2)), lin.weight, lin.bias, ACT_NONE)
    print("dec_linear2:", d(unframes(got, 80), y.transpose(1, 2)))
    rec = y.transpose(1, 2)
    ref = m.postnet_fwd(rec)
    got = hm.postnet.forward_frames(frames(rec), B, 1, residual=None)
    print("postnet:", d(unframes(got, 80), ref))
