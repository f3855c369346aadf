#!/usr/bin/env python3
"""Generate include/osp.h from the extern "C" definitions in optispeech_amd/csrc (keeps the header and the
library in lock-step; tests/test_abi.py checks that every declared symbol is exported)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optispeech_amd", "csrc")

# symbol -> reference interface it replaces (paths relative to mush42/optispeech @ 2024-12-20)
REPLACES = {
    "osp_conv_gemm_f32": "nn.Linear / nn.Conv1d forward + dgrad: generator/modules/convnext.py:39-41, modules/core.py:66-71,95, "
                         "generator/alignments.py:55-64, vocoder/wavenext/__init__.py:43-44,83; torch.matmul in alignments.py:173",
    "osp_conv_gemm_f32_split": "same call sites as osp_conv_gemm_f32 (f32 operands in HBM), products as three bf16 MFMAs over (hi, lo) "
                               "operand pairs: the 'mixed' parity mode's generator GEMMs outside the index-critical path",
    "osp_conv_wgrad_f32": "autograd weight/bias gradients of the same nn.Linear / nn.Conv1d call sites",
    "osp_conv_gemm_bf16": "same call sites as osp_conv_gemm_f32 and the weight-normed Conv2d (k,1) stacks of DiscriminatorP "
                          "(vocoder/wavenext/disc/_discriminators.py:51-60,80-90), bf16 operands / f32 accumulate",
    "osp_conv_wgrad_bf16": "weight gradients of the same, bf16 operands / f32 accumulate",
    "osp_conv_wgrad_bf16_ws": "the same weight gradients with a caller-supplied split workspace: no f32 atomics, bit-reproducible "
                              "(autograd of nn.Conv1d / nn.Linear: generator/modules/convnext.py:39-41)",
    "osp_conv_wgrad_f32_ws": "exact-f32 weight gradients (the f32 / mixed parity modes) with a split workspace: no f32 atomics "
                             "(autograd of nn.Conv1d / nn.Linear: generator/modules/convnext.py:39-41, generator/alignments.py:55-64)",
    "osp_conv_wgrad_f32_split_ws": "the same weight gradients (f32 operands in HBM) with products as three bf16 MFMAs over (hi, lo) operand "
                                   "pairs: the 'mixed' parity mode's generator",
    "osp_conv2d_wgrad_bf16_ws": "autograd weight gradients of the DiscriminatorP / DiscriminatorR Conv2d stacks "
                                "(vocoder/wavenext/disc/_discriminators.py:51-60,154-163) with a split workspace instead of atomics",
    "osp_dwconv7_ln_fwd": "ConvNeXtBlock.forward dwconv + LayerNorm: generator/modules/convnext.py:36-38",
    "osp_dwconv7_bwd": "autograd of nn.Conv1d(groups=dim) at generator/modules/convnext.py:36",
    "osp_layernorm_fwd": "nn.LayerNorm: convnext.py:85,102; modules/layers.py:26-45 (eps 1e-12) + nn.Dropout core.py:74; wavenext/__init__.py:84",
    "osp_layernorm_bwd": "autograd of the LayerNorm (+ReLU, +Dropout) call sites above",
    "osp_text_embed_fwd": "TextEmbedding.forward modules/core.py:25-31, ScaledSinusoidalEmbedding modules/layers.py:48-71",
    "osp_text_embed_bwd": "autograd of nn.Embedding(padding_idx=0) + ScaledSinusoidalEmbedding.scale",
    "osp_lgamma_table": "scipy.special gammaln inside scipy.stats.betabinom.logpmf (alignments.py:114)",
    "osp_betabinom_prior": "AlignmentModule._generate_prior generator/alignments.py:85-123",
    "osp_pairwise_score": "AlignmentModule.forward generator/alignments.py:66-72",
    "osp_logsoftmax_prior_fwd": "AlignmentModule.forward generator/alignments.py:74-81",
    "osp_logsoftmax_prior_bwd": "autograd of generator/alignments.py:66-81",
    "osp_mas_workspace_bytes": "workspace query for osp_mas",
    "osp_mas": "_monotonic_alignment_search + viterbi_decode generator/alignments.py:177-239 (numba JIT + np.bincount)",
    "osp_bin_loss_bwd": "autograd of `bin_loss - cur_log_p_attn[t_idx, viterbi].mean()` generator/alignments.py:236-238",
    "osp_duration_stats": "_average_by_duration / average_by_duration generator/alignments.py:242-280; cumsum at :167",
    "osp_gaussian_weights": "GaussianUpsampling.forward generator/alignments.py:163-172",
    "osp_gather_rows": "get_segments / get_segments_numpy utils/segments.py:41-72 (callers generator/__init__.py:147-158, base_lightning_module.py:38-44)",
    "osp_expand_by_duration": "expand_by_duration generator/alignments.py:283-297",
    "osp_variance_losses": "FastSpeech2Loss.forward generator/loss.py:83-140, DurationPredictorLoss :28-46",
    "osp_forwardsum_ctc_workspace_floats": "workspace query for osp_forwardsum_ctc",
    "osp_forwardsum_ctc": "ForwardSumLoss.forward generator/loss.py:150-194 (F.log_softmax + F.ctc_loss per item) and its autograd",
    "osp_fft_twiddles": "twiddle table for the STFT kernels (torch.stft internals)",
    "osp_stft_mag_fwd": "torch.stft(...).abs(): disc/loss.py:123-142, disc/_discriminators.py:196-216, torchaudio MelSpectrogram disc/loss.py:94-107",
    "osp_stft_mag_bwd": "autograd of the same",
    "osp_sumsq": "clip_grad_norm_ total norm (LightningModule.clip_gradients, base_lightning_module.py:100-102,120-122)",
    "osp_adamw_clip": "torch.optim.AdamW.step + clip_gradients: base_lightning_module.py:96-105,116-125; configs/model/optimizer/adamw.yaml",
    "osp_attn_softmax_fwd": "MultiHeadedAttention.forward_attention masked softmax + dropout: generator/modules/_transformer/attention.py:75-97",
    "osp_attn_softmax_bwd": "autograd of the same (softmax' and the dropout mask regenerated from the Philox counter)",
    "osp_attn_train_fwd": "MultiHeadedAttention.forward scores -> masked softmax -> dropout -> P V in ONE kernel (training): "
                          "generator/modules/_transformer/attention.py:80-98,120-125; keeps the per-row log-sum-exp only",
    "osp_attn_train_bwd": "autograd of the same: probabilities recomputed per tile from q, k and the log-sum-exp (no (T x T) tensor)",
    "osp_conv2d_gemm_bf16": "weight-normed Conv2d forward of DiscriminatorP / DiscriminatorR + F.leaky_relu: "
                            "vocoder/wavenext/disc/_discriminators.py:51-60,63-97 and :154-163,165-194",
    "osp_conv2d_dgrad_bf16": "autograd input gradients of the same strided Conv2d stacks (all stride phases in one launch)",
    "osp_conv2d_wgrad_bf16": "autograd weight gradients of the same Conv2d stacks",
    "osp_smallcin_conv_fwd": "first layers Conv2d(1, 32, (k,1)) / Conv2d(1, 32, (7,5)) and the 32-channel DiscriminatorR layers: "
                             "disc/_discriminators.py:53,156-160",
    "osp_smallcin_conv_wgrad": "autograd weight/bias gradients of the same narrow-channel layers",
    "osp_logmel_energy": "CommonFeatureExtractor.get_mel mel_basis @ magnitudes + spectral_normalize_torch, and get_energy's torch.norm: "
                         "dataset/feature_extractors/__init__.py:143-146,197-199; utils/audio.py:23-24",
    "osp_cast_bf16": "no reference counterpart: f32 -> bf16 operand copy for the MFMA kernels (the reference's autocast does this implicitly)",
    "osp_pack_bf16": "no reference counterpart: (Cout, Cin, K) -> tap-major bf16 weight layout for the conv-GEMM kernels",
    "osp_wnorm_fwd": "torch.nn.utils.weight_norm forward w = g * v / ||v|| at every DiscriminatorP/R conv: disc/_discriminators.py:7,53-60,156-163",
    "osp_wnorm_bwd": "autograd of weight_norm: gradients of weight_g / weight_v from the effective-weight gradient",
    "osp_l1_sum": "FeatureMatchingLoss.forward torch.mean(torch.abs(rl - gl)): disc/loss.py:68-85",
    "osp_l1_sign": "autograd of the same",
    "osp_hinge_sum": "GeneratorLoss / DiscriminatorLoss hinge terms mean(clamp(1 -/+ d, min=0)): disc/loss.py:11-65",
    "osp_hinge_grad": "autograd of the same",
    "osp_l1_sum_multi": "FeatureMatchingLoss.forward over ALL feature-map pairs of a discriminator family in one launch: disc/loss.py:68-85",
    "osp_l1_sign_multi": "autograd of the same",
    "osp_hinge_sum_multi": "GeneratorLoss / DiscriminatorLoss hinge terms of all sub-discriminators in one launch: disc/loss.py:11-65",
    "osp_hinge_grad_multi": "autograd of the same",
    "osp_wnorm_fwd_multi": "torch.nn.utils.weight_norm forward of many convs in one launch (see osp_wnorm_fwd)",
    "osp_wnorm_bwd_multi": "autograd of weight_norm for many convs in one launch (see osp_wnorm_bwd)",
    "osp_colsum_prod": "autograd of the layer scale `self.gamma * x`: ConvNeXtBlock.forward generator/modules/convnext.py:45-46",
    "osp_cast_bf16_rows": "no reference counterpart: row-scaled f32 -> bf16 operand copy (drop-path / mask factor folded in)",
    "osp_conv1d_dilated_fwd": "CausalConv1d / nn.Conv1d(dilation=d): vocoder/streaming_hifigan/modules/conv_layer.py:18-60,118-159 (callers residual_block.py:44-70)",
    "osp_conv1d_dilated_bwd": "autograd of the same (input, weight and bias gradients)",
    "osp_conv_transpose1d_fwd": "CausalConvTranspose1d / nn.ConvTranspose1d(stride s): vocoder/streaming_hifigan/modules/conv_layer.py:63-115,162-200",
    "osp_conv_transpose1d_bwd": "autograd of the same (input and weight gradients)",
    "osp_comm_unique_id": "rendezvous id of the gradient communicator (Lightning DDP strategy setup, configs/trainer/ddp.yaml:4-9)",
    "osp_comm_init": "DDP process-group construction for the gradient all-reduce (configs/trainer/ddp.yaml:4-9)",
    "osp_comm_world": "number of ranks of the initialised communicator (0 = none)",
    "osp_allreduce_bucket": "DDP's bucketed gradient all-reduce (torch.nn.parallel.DistributedDataParallel reducer behind manual_backward, base_lightning_module.py:99,119)",
    "osp_comm_destroy": "process-group teardown",
    "osp_spectral_loss_sums": "SpectralConvergenceLoss + LogSTFTMagnitudeLoss reductions (torch.norm / F.l1_loss of log magnitudes) and the mel L1: disc/loss.py:107-120,231-270",
    "osp_spectral_loss_bwd": "autograd of the same with respect to the predicted magnitudes",
    "osp_pack_bf16_multi": "no reference counterpart: osp_pack_bf16 for a list of weights in one launch",
    "osp_lstm_fwd": "recurrence of nn.LSTM(dim, dim, 1, batch_first=True) after the input-projection GEMM: generator/modules/leanspeech.py:49-60 (hx / flags: workspace, flags zeroed)",
    "osp_lstm_bwd": "autograd of the same recurrence: gradient w.r.t. the gate pre-activations for every step (dW / db / dx are GEMMs over it)",
    "osp_pack_bf16_kperm16": "no reference counterpart: bf16 weight pack of pwconv2 in the k order osp_convnext_mlp_fused reads it in",
    "osp_convnext_mlp_fused": "ConvNeXtBlock.forward pwconv1 -> GELU -> pwconv2 -> gamma, residual (+ backbone mask) without gradients, hidden "
                              "activations never in HBM: generator/modules/convnext.py:39-46,99-101 (callers: OptiSpeechGenerator.synthesise "
                              "generator/__init__.py:170-228, WaveNeXt.forward vocoder/wavenext/__init__.py:77-88)",
    "osp_convnext_mlp_fused_live": "the same, walking the rows in a given order (unmasked rows first) with the caller's count of unmasked rows (synthesise knows the "
                                   "sum of the utterance lengths after its one length sync, generator/__init__.py:258-262): masked row blocks leave at once",
    "osp_row_order": "no reference counterpart: the rows of a padded batch in the order 'unmasked first' (a stable partition by the padding mask), "
                     "which osp_convnext_mlp_fused_live walks",
    "osp_attn_fused_fwd": "MultiHeadedAttention.forward_attention without the (T x T) scores in HBM (no-grad / inference path): _transformer/attention.py:75-101",
    "osp_dwconv_fwd": "depthwise nn.Conv1d(groups = C, odd k) and its input gradient (flip = 1): ConvSeparable modules/layers.py:455-477, _conformer/convolution.py",
    "osp_dwconv_wgrad": "autograd weight / bias gradient of the same depthwise Conv1d",
    "osp_dropout_add": "F.dropout (+ residual add) call sites of the separable-conv layers: modules/layers.py:497-503; backward = the same kernel on the gradient",
    "osp_ln_dwconv7_bwd": "autograd of LayerNorm + depthwise Conv1d(k=7) of ConvNeXtBlock in one pass: generator/modules/convnext.py:36-38 (C <= 256)",
    "osp_period_fold": "DiscriminatorP.forward reflect pad + (b, t/p, p) view as period-column sequences, and its gradient (backward = 1): vocoder/wavenext/disc/_discriminators.py:63-72",
    "osp_drop_path_rows": "DropPath factors of all blocks of a ConvNeXt backbone: generator/modules/convnext.py:121-129 (drop_p_host: plain host array)",
    "osp_segment_starts": "get_random_segments start indices: utils/segments.py:12-38 (caller generator/__init__.py:147-153)",
    "osp_last_error": "error text of the last failing call on this thread",
    "osp_abi_version": "ABI version of this library",
    "osp_kernel_note_host": "measurement aid, no reference counterpart: symbol and algorithmic flops of the matrix-core kernel(s) the calling "
                            "thread's last entry-point call launched (bench.py's roofline block reads it; cleared by the read)",
    "osp_kernel_note_bytes_host": "measurement aid: algorithmic HBM bytes of the launches osp_kernel_note_host reports (decides matrix-pipe-bound vs HBM-bound)",
    "osp_clip": "torch.clip(audio, -1, 1) of WaveNeXtHead.forward and its backward: vocoder/wavenext/__init__.py:47",
    "osp_stream_handover": "torch.cuda.Event.record + Stream.wait_event of the multi-stream schedule in one call (no reference counterpart: "
                           "the reference runs on one stream)",
    "osp_memset": "no reference counterpart: zero / byte fill of a buffer inside a taped region (torch.zeros / Tensor.zero_ of the host code)",
    "osp_copy": "no reference counterpart: device-to-device copy inside a taped region (torch.cat / Tensor.copy_ of the host code)",
    "osp_tape_selftest": "test aid of the call tapes: adds into a HOST counter, no device work (lets the CPU suite record / patch / replay)",
    "osp_tape_selftest_table": "test aid of the call tapes: sums host int64 values addressed through a host table (copy + patching of descriptor tables)",
    "osp_store_i64": "no reference counterpart: the per-step dropout seed written to device memory (kernels read it through seed_dev)",
    "osp_ew_axpby": "autograd's gradient accumulation (a tensor with several consumers), negation and scalar scaling inside taped regions",
    "osp_ew_mul": "element-wise products with a row / column broadcast: `x * mask` sites of modules/core.py:161-175 and their autograd",
    "osp_ew_scale_dev": "autograd of the scalar loss assembly: gradient tensor times an incoming device scalar (generator/loss.py, alignments.py:236-238)",
    "osp_ew_relu_mask": "autograd of nn.ReLU after nn.Conv1d: modules/core.py:66-71",
    "osp_transpose_last2": "mel.transpose(1, 2) of OptiSpeechGenerator.forward: generator/__init__.py:122",
    "osp_length_masks": "sequence_mask + padding masks: utils/model.py:12-16, generator/__init__.py:96-103",
    "osp_posenc_fwd": "ScaledPositionalEncoding.forward `x + alpha * pe`: generator/modules/_transformer/embedding.py:120-124",
    "osp_posenc_dalpha": "autograd of the same w.r.t. alpha (stage 1: one partial per workgroup; stage 2 = osp_sum_scaled)",
    "osp_permute_0213": "the head split / merge `.view(B, T, h, d_k).transpose(1, 2)` (+ .contiguous()) of MultiHeadedAttention: "
                        "generator/modules/_transformer/attention.py:59-66,99-101",
    "osp_sum_scaled": "torch.mean over the per-utterance loss terms: generator/loss.py:190-193, alignments.py:236-238",
    "osp_dot_multi": "weighted loss sums: generator/__init__.py:175-181, vocoder/wavenext/disc/__init__.py:105-111",
    "osp_scale_vec": "autograd of the weighted loss sums",
    "osp_source_hash": "content hash of the sources this library was built from (optispeech_amd/build.py checks it; no reference counterpart)",
}

PREAMBLE = '''/* libosp_hip -- C ABI of the MI355X-native OptiSpeech hot path (generated by tools/gen_header.py; do not edit).
 *
 * Conventions
 *   - every entry point is extern "C", returns int (0 = OSP_OK, < 0 = error; text via osp_last_error()),
 *     never throws, never allocates, never synchronises: work is enqueued on the caller's hipStream_t;
 *     `*_workspace_*` queries return sizes and take no stream
 *   - pointers are DEVICE pointers; activations are channels-last (B, T, C) contiguous f32; utterance lengths are
 *     int64 device arrays; sizes are int64_t, real scalars float
 *   - parameter gradients are ACCUMULATED (+=, f32 atomics) into the caller's gradient arena
 *   - "replaces:" names the reference interface (file:line in mush42/optispeech @ 2024-12-20) each call stands in for
 */
#ifndef OSP_H
#define OSP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ihipStream_t* hipStream_t;

#define OSP_OK 0
#define OSP_ERR_ARG (-1)
#define OSP_ERR_HIP (-2)
#define OSP_ERR_UNSUPPORTED (-3)

/* epilogues of osp_conv_gemm_* */
enum { OSP_EPI_NONE = 0, OSP_EPI_RELU = 1, OSP_EPI_GELU = 2, OSP_EPI_SCALE_RES_MASK = 3, OSP_EPI_GELU_BWD = 4,
       OSP_EPI_RELU_BWD = 5, OSP_EPI_AXMY = 6, OSP_EPI_MASK = 7, OSP_EPI_LRELU = 8, OSP_EPI_LRELU_BWD = 9 };

'''


def main():
    protos = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".cpp")):
            continue
        src = open(os.path.join(CSRC, f)).read()
        for m in re.finditer(r'extern "C"\s+([^{;]+?)\s*\{', src):
            sig = " ".join(m.group(1).split())
            name = re.search(r"(osp_\w+)\s*\(", sig).group(1)
            protos.append((f, name, sig))
    out = [PREAMBLE]
    cur = None
    for f, name, sig in protos:
        if f != cur:
            out.append(f"/* ---- {f} ---- */\n")
            cur = f
        out.append(f"/* replaces: {REPLACES[name]} */\n{sig};\n\n")
    out.append("#ifdef __cplusplus\n}\n#endif\n#endif /* OSP_H */\n")
    os.makedirs(os.path.join(ROOT, "include"), exist_ok=True)
    with open(os.path.join(ROOT, "include", "osp.h"), "w") as fh:
        fh.write("".join(out))
    print(f"include/osp.h: {len(protos)} entry points")


if __name__ == "__main__":
    main()
