"""Oracle (test infrastructure): alignment module, MAS, duration averaging, length regulators.

All citations: optispeech/model/generator/alignments.py unless stated otherwise.
"""
import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

from .nn_ops import conv1d_cl

_HERE = os.path.dirname(os.path.abspath(__file__))
_CLIB = None


def build_clib(force=False):
    """gcc-compile oracle/mas.c -> oracle/_build/libosp_oracle.so (checker only)."""
    out_dir = os.path.join(_HERE, "_build")
    so = os.path.join(out_dir, "libosp_oracle.so")
    src = os.path.join(_HERE, "mas.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", so, src, "-lm"])
    return so


def clib():
    global _CLIB
    if _CLIB is None:
        _CLIB = ctypes.CDLL(build_clib())
        _CLIB.osp_oracle_mas.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _CLIB.osp_oracle_avg_by_dur.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 3 + [ctypes.c_void_p]
    return _CLIB


# --------------------------------------------------------------------------- A7 / A7b
def betabinom_prior_np(T, N, w=1.0):
    """(T,N) float64 log prior for one item; _generate_prior :110-114 (scipy betabinom.logpmf)."""
    from scipy.stats import betabinom
    alpha = w * np.arange(1, T + 1, dtype=float)
    beta = w * np.array([T - t + 1 for t in alpha])
    k = np.arange(N)[:, None]
    return betabinom.logpmf(k, N, alpha, beta).T            # :114 then :121 transpose -> (T,N)


def betabinom_prior_lgamma_np(T, N):
    """Same quantity from log-factorials only (all arguments are integers for w=1):
    logpmf(k; n=N, a=t, b=T-t+1) = lchoose(N,k) + lbeta(k+a, N-k+b) - lbeta(a,b).
    This is the closed form the device kernel evaluates; checked against scipy in tests."""
    lg = np.array([math.lgamma(i) if i > 0 else 0.0 for i in range(T + N + 4)], dtype=np.float64)

    def lbeta(x, y):
        return lg[x] + lg[y] - lg[x + y]

    t = np.arange(1, T + 1)[:, None]      # a
    b = T - t + 1
    k = np.arange(N)[None, :]
    lchoose = lg[N + 1] - lg[k + 1] - lg[N - k + 1]
    return lchoose + lbeta(k + t, N - k + b) - lbeta(t, b)


def batched_prior(text_lengths, feats_lengths, T_feats=None, T_text=None):
    """_generate_prior :85-123 -> (B,T_feats,T_text) float32 with -inf outside the valid block."""
    B = len(text_lengths)
    T_text = int(text_lengths.max()) if T_text is None else T_text
    T_feats = int(feats_lengths.max()) if T_feats is None else T_feats
    out = torch.full((B, T_feats, T_text), -np.inf)                       # :103
    for b in range(B):
        T, N = int(feats_lengths[b]), int(text_lengths[b])
        out[b, :T, :N] = torch.from_numpy(betabinom_prior_np(T, N))      # :121-122 (float64 -> float32 on assign)
    return out


def alignment_features(text, feats, P, pre):
    """Conv stacks of AlignmentModule.forward :55-64. text (B,Tt,adim), feats (B,Tf,odim) channels-last."""
    t = F.relu(conv1d_cl(text, P[pre + "t_conv1.weight"], P[pre + "t_conv1.bias"], 1))
    t = conv1d_cl(t, P[pre + "t_conv2.weight"], P[pre + "t_conv2.bias"], 0)
    f = F.relu(conv1d_cl(feats, P[pre + "f_conv1.weight"], P[pre + "f_conv1.bias"], 1))
    f = F.relu(conv1d_cl(f, P[pre + "f_conv2.weight"], P[pre + "f_conv2.bias"], 1))
    f = conv1d_cl(f, P[pre + "f_conv3.weight"], P[pre + "f_conv3.bias"], 0)
    return t, f


def pairwise_logprob(t, f, x_pad_mask, prior, chunk=64):
    """:66-81: score = -||f_t - e_n||_2, masked_fill(-inf) on padded text, log_softmax over text, + prior.

    Chunked over frames so the (B,Tf,Tt,C) difference tensor of the reference (:66) is never
    materialised at BASELINE size; the arithmetic per element is identical.
    """
    outs = []
    for s in range(0, f.shape[1], chunk):
        d = f[:, s:s + chunk, None, :] - t[:, None, :, :]
        outs.append(-torch.norm(d, p=2, dim=3))
    score = torch.cat(outs, dim=1)
    if x_pad_mask is not None:
        score = score.masked_fill(x_pad_mask[:, None, :], -np.inf)        # :70-72
    return F.log_softmax(score, dim=-1) + prior                           # :74-81


def alignment_logprob(text, feats, text_lengths, feats_lengths, x_pad_mask, P, pre):
    """AlignmentModule.forward :41-83."""
    t, f = alignment_features(text, feats, P, pre)
    prior = batched_prior(text_lengths, feats_lengths, feats.shape[1], text.shape[1])
    return pairwise_logprob(t, f, x_pad_mask, prior)


# --------------------------------------------------------------------------- A8
def mas_path_np(lp):
    """_monotonic_alignment_search :177-207 in numpy. lp (T_mel,T_inp) float32 -> int64 (T_mel,).

    Row 0 follows the numba semantics assumed in oracle/mas.c (sequential float32 prefix sum).
    """
    T_mel, T_inp = lp.shape
    Q = np.full((T_inp, T_mel), -np.inf)
    logp = lp.T
    Q[0] = np.cumsum(logp[0], dtype=np.float32).astype(np.float64)
    for j in range(1, T_mel):
        hi = min(j + 1, T_inp)
        if hi > 1:
            Q[1:hi, j] = np.maximum(Q[0:hi - 1, j - 1], Q[1:hi, j - 1]) + logp[1:hi, j].astype(np.float64)
    A = np.full((T_mel,), T_inp - 1, dtype=np.int64)
    for j in range(T_mel - 2, -1, -1):
        i_b = A[j + 1]
        i_a = i_b - 1
        if i_b == 0:
            A[j] = 0
        elif Q[i_a, j] >= Q[i_b, j]:
            A[j] = i_a
        else:
            A[j] = i_b
    return A


def mas_path_c(lp):
    lp = np.ascontiguousarray(lp, dtype=np.float32)
    T_mel, T_inp = lp.shape
    path = np.empty((T_mel,), dtype=np.int64)
    rc = clib().osp_oracle_mas(lp.ctypes.data, T_mel, T_inp, T_inp, path.ctypes.data)
    assert rc == 0
    return path


def viterbi_decode(log_p_attn, text_lengths, feats_lengths, use_c=True):
    """viterbi_decode :210-239 -> (ds (B,T_text) float32, bin_loss scalar, paths list[int64 arrays])."""
    B, _, T_text = log_p_attn.shape
    ds = torch.zeros((B, T_text))
    bin_loss = 0
    paths = []
    for b in range(B):
        cur = log_p_attn[b, : int(feats_lengths[b]), : int(text_lengths[b])]
        lp = cur.detach().float().cpu().numpy()
        path = mas_path_c(lp) if use_c else mas_path_np(lp)
        paths.append(path)
        cnt = np.bincount(path)
        ds[b, : len(cnt)] = torch.from_numpy(cnt).float()
        t_idx = torch.arange(int(feats_lengths[b]))
        bin_loss = bin_loss - cur[t_idx, torch.from_numpy(path)].mean()   # :237
    return ds, bin_loss / B, paths


# --------------------------------------------------------------------------- A9
def average_by_duration(ds, xs, text_lengths, feats_lengths):
    """average_by_duration/_average_by_duration :242-280. ds (B,Tt) float, xs (B,Tf) -> (B,Tt) float32."""
    B, Tt = ds.shape
    Tf = xs.shape[1]
    d = np.ascontiguousarray(ds.detach().float().numpy())
    x = np.ascontiguousarray(xs.detach().float().numpy())
    tl = np.ascontiguousarray(text_lengths.numpy().astype(np.int64))
    fl = np.ascontiguousarray(feats_lengths.numpy().astype(np.int64))
    out = np.zeros((B, Tt), dtype=np.float32)
    clib().osp_oracle_avg_by_dur(d.ctypes.data, x.ctypes.data, tl.ctypes.data, fl.ctypes.data, B, Tt, Tf,
                                 out.ctypes.data)
    return torch.from_numpy(out)


# --------------------------------------------------------------------------- A10
def gaussian_upsampling(hs, ds, h_masks, d_masks, delta=0.1):
    """GaussianUpsampling.forward :136-174. hs (B,Tt,C), ds (B,Tt); masks True=valid."""
    B = ds.shape[0]
    ds = ds.clone()
    if ds.sum() == 0:                                                   # :152-157
        ds[ds.sum(dim=1).eq(0)] = 1
    T_feats = h_masks.shape[-1]
    t = torch.arange(0, T_feats)[None, :].repeat(B, 1).float() * h_masks.float()   # :163-165
    c = ds.cumsum(dim=-1) - ds / 2                                                  # :167
    energy = -1 * delta * (t[:, :, None] - c[:, None, :]) ** 2                      # :168
    energy = energy.masked_fill(~d_masks[:, None, :].expand(-1, T_feats, -1), -float("inf"))   # :170
    p = torch.softmax(energy, dim=2)                                                # :172
    return torch.matmul(p, hs)                                                      # :173


def expand_by_duration(x, durations):
    """expand_by_duration :283-297 as an index gather. x (B,Tt,C), durations int64 (B,Tt)."""
    lengths = durations.sum(dim=1)
    max_len = int(lengths.max())
    cum = torch.cumsum(F.pad(durations, (1, 0)), dim=1)                  # (B,Tt+1)
    r = torch.arange(max_len)[None, :, None]
    hit = (cum[:, None, :-1] <= r) & (cum[:, None, 1:] > r)              # (B,max_len,Tt) one-hot or empty
    return torch.matmul(hit.to(x.dtype), x), lengths


# --------------------------------------------------------------------------- A10c
def segment_starts(lengths, segment_size, rand01):
    """get_random_segments start indices, utils/segments.py:29-34, with the uniform draws injected."""
    max_start = (lengths - segment_size).clamp(min=0)
    return (rand01 * max_start).to(torch.long)


def gather_segments(x, starts, segment_size):
    """get_segments utils/segments.py:41-60 on channels-last x (B,T,C) -> (B,segment_size,C)."""
    return torch.stack([x[i, int(s): int(s) + segment_size] for i, s in enumerate(starts)])
