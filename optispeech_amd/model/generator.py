"""Host-side mirror of optispeech/model/generator/__init__.py (OptiSpeechGenerator).

Same constructor (partials for the sub-modules), same ``forward`` / ``synthesise`` signatures and output
dicts.  Differences that are invisible at the interface: activations stay channels-last, and the
alignment search / duration averaging / segment slicing run on the device, so a training step has no
host round trip (the reference has >= 2B+4).
"""
from time import perf_counter

import torch
from torch import nn

from .. import kernels as K
from .. import ops, precision
from .alignments import (AlignmentModule, GaussianUpsampling, average_by_duration, expand_by_duration,
                         viterbi_decode)


def padding_mask(lengths, T):
    """~sequence_mask(lengths, T) -> (B, T) bool, True = padding; on the GPU one launch yields it together with the f32 keep mask
    every module of the forward asks for (modules.row_mask finds it on the tensor)."""
    if not lengths.is_cuda:
        return ~sequence_mask(lengths, T)
    pad, keep = K.length_masks(lengths.contiguous(), int(T))
    try:
        pad._osp_rowmask = ((-1 if pad.is_inference() else pad._version, torch.cuda.is_current_stream_capturing()), keep)
        pad._osp_lengths = (-1 if pad.is_inference() else pad._version, lengths)      # the Transformer backbone's valid-key counts
    except (AttributeError, RuntimeError):
        pass
    return pad


def sequence_mask(length, max_length=None):
    """utils/model.py:12-16."""
    if max_length is None:
        max_length = length.max()
    x = torch.arange(max_length, dtype=length.dtype, device=length.device)
    return x.unsqueeze(0) < length.unsqueeze(1)


_VOC_STREAM = __import__("os").environ.get("OSP_VOC_STREAM", "1") != "0"


class OptiSpeechGenerator(nn.Module):
    def __init__(self, dim: int, segment_size, text_embedding, encoder, duration_predictor, pitch_predictor,
                 energy_predictor, decoder, vocoder, loss_coeffs, feature_extractor, num_speakers, num_languages,
                 data_statistics, **kwargs):
        super().__init__()
        self.segment_size = segment_size
        self.loss_coeffs = loss_coeffs
        self.n_feats = feature_extractor.n_feats
        self.n_fft = feature_extractor.n_fft
        self.hop_length = feature_extractor.hop_length
        self.sample_rate = feature_extractor.sample_rate
        self.data_statistics = data_statistics
        self.num_speakers = num_speakers
        self.num_languages = num_languages

        self.text_embedding = text_embedding(dim=dim)
        self.encoder = encoder(dim=dim)
        self.duration_predictor = duration_predictor(dim=dim)
        self.alignment_module = AlignmentModule(adim=dim, odim=self.n_feats)
        self.pitch_predictor = pitch_predictor(dim=dim)
        self.energy_predictor = energy_predictor(dim=dim)
        self.feature_upsampler = GaussianUpsampling()
        self.decoder = decoder(dim=dim)
        self.vocoder = vocoder(input_channels=dim, sample_rate=self.sample_rate, n_fft=self.n_fft,
                               hop_length=self.hop_length)
        if self.num_speakers > 1:
            self.sid_embed = torch.nn.Embedding(self.num_speakers, dim)
        if self.num_languages > 1:
            self.lid_embed = torch.nn.Embedding(self.num_languages, dim)
        #: test hook: fixed uniform draws in [0,1) for the segment starts (None = torch.rand)
        self.segment_rand01 = None
        #: synthesise(): replay the shape-static part (upsampler, decoder, vocoder) from captured hipGraphs
        self.graph_decode = __import__("os").environ.get("OSP_GRAPH_DECODE", "0") == "1"
        self._decode_graphs = {}
        #: with graph_decode: the shape-static part BEFORE the length sync (text encoder, predictors, length sums) replays from a
        #: captured hipGraph as well (OSP_GRAPH_ENCODE=0: eager launches there, the round-4 schedule)
        self.graph_encode = __import__("os").environ.get("OSP_GRAPH_ENCODE", "1") == "1"
        self._encode_graphs = {}

    # ------------------------------------------------------------------------------------------ training forward
    def forward(self, x, x_lengths, mel, mel_lengths, pitches, energies, sids, lids):
        """generator/__init__.py:72-192.  ``mel`` arrives in the reference layout (B, n_feats, T_mel)."""
        return self._forward_am(x, x_lengths, mel, mel_lengths, pitches, energies, sids, lids, vocoder_hook=self._run_vocoder)

    def _run_vocoder(self, segment):
        # :161 (f0 unused by WaveNeXt).  The vocoder's graph is disjoint from the acoustic model's (``segment`` is detached):
        # built on its own stream, its backward overlaps the acoustic model's backward
        if _VOC_STREAM and segment.is_cuda and torch.is_grad_enabled():
            return ops.run_on_side_stream("vocoder", lambda: self.vocoder(segment, f0=None), [segment])
        return self.vocoder(segment, f0=None)

    def draw_segment_rand(self, B, device):
        """The uniform draws of get_random_segments (utils/segments.py:29-34) as a contiguous f32 device vector: the test hook
        ``segment_rand01`` or torch.rand.  Drawn OUTSIDE the taped acoustic-model segment (torch's generator is host state)."""
        r = self.segment_rand01 if self.segment_rand01 is not None else torch.rand(B, device=device)
        return r.to(device=device, dtype=torch.float32).contiguous()

    def _forward_am(self, x, x_lengths, mel, mel_lengths, pitches, energies, sids, lids, vocoder_hook=None, rand01=None):
        """The acoustic-model part of forward(): everything up to the detached decoder segment, plus the acoustic losses (which
        do not depend on the vocoder).  ``vocoder_hook(segment) -> wav_hat`` runs where the reference calls the vocoder (:161);
        None = the caller runs the vocoder itself (graph-segment mode, optispeech_amd/graphs.py)."""
        B = x.shape[0]
        Tt, Tm = x.shape[1], mel.shape[2]
        x_lengths = x_lengths.contiguous()
        mel_lengths = mel_lengths.contiguous()
        input_padding_mask = padding_mask(x_lengths, Tt)                    # :96-102
        target_padding_mask = padding_mask(mel_lengths, Tm)                 # :99-103

        with precision.index_path():        # exact-f32 forward of everything the (discrete) alignment depends on, in every mode
            h, _ = self.text_embedding(x)                                       # :106
            h = self.encoder(h, input_padding_mask)                             # :109
            if sids is not None:
                h = h + self.sid_embed(sids.view(-1)).unsqueeze(1)              # :112-114
            if lids is not None:
                h = h + self.lid_embed(lids.view(-1)).unsqueeze(1)              # :115-117

            feats = K.transpose_last2(mel.contiguous()) if mel.is_cuda else mel.transpose(1, 2).contiguous()   # :122
            log_p_attn = self.alignment_module(text=h, feats=feats, text_lengths=x_lengths, feats_lengths=mel_lengths,
                                               x_masks=input_padding_mask)      # :120-126
            durations, path, bin_item = viterbi_decode(log_p_attn, x_lengths, mel_lengths)      # :127
        # :174 -- issued here (side stream) so that it overlaps everything up to the loss sum; joined below
        forwardsum_loss, bin_loss = ops.AlignLossFn.apply(log_p_attn, x_lengths, mel_lengths, path, bin_item)
        duration_hat = self.duration_predictor(h.detach(), input_padding_mask)              # :128
        p_avg, e_avg = average_by_duration(durations, pitches, energies, x_lengths, mel_lengths)   # :131-132
        h, pitch_hat = self.pitch_predictor(h, input_padding_mask, p_avg)                   # :135
        h, energy_hat = self.energy_predictor(h, input_padding_mask, e_avg)                 # :136
        y = self.feature_upsampler(h, durations, x_lengths, mel_lengths, Tm)                # :139-141
        # :144 -- only the DETACHED decoder output is used below (:149-161: the vocoder sees segment.detach(), and nothing else
        # reads y), so the decoder receives no gradient in the reference either: run it without a tape (no saved activations)
        with torch.no_grad():
            y = self.decoder(y.detach(), target_padding_mask)

        segment_size = min(self.segment_size, y.shape[1])                                   # :147
        r = rand01 if rand01 is not None else self.draw_segment_rand(B, y.device)
        # :148 + utils/segments.py:29-34 in one launch: long(r * clamp(float(len - 4) - segment_size, 0))
        start_idx = K.segment_starts(r, mel_lengths, segment_size)
        segment = K.gather_rows(y.detach(), start_idx, segment_size)                        # :149-153, detach :161
        wav_hat = vocoder_hook(segment) if vocoder_hook is not None else None

        c = self.loss_coeffs
        duration_loss, pitch_loss, energy_loss = ops.VarianceLossFn.apply(
            duration_hat, pitch_hat, energy_hat, durations, p_avg, e_avg, x_lengths)        # :165-173
        ops.join_side_stream()                                                              # the CTC recursion ran alongside
        align_loss = forwardsum_loss.detach() + bin_loss.detach()                           # :175 (logged value)
        # :176-181 as one node: lambda_align * (forwardsum + bin) + lambda_d * dur + lambda_p * pitch + lambda_e * energy
        loss = ops.weighted_sum([forwardsum_loss, bin_loss, duration_loss, pitch_loss, energy_loss],
                                [c.lambda_align, c.lambda_align, c.lambda_duration, c.lambda_pitch, c.lambda_energy])
        # NB: the reference moves the sub-losses to the CPU here (4 device syncs); we keep them on the device and
        # let the caller fetch all scalars with one copy.
        return {"wav_hat": wav_hat, "start_idx": start_idx, "segment_size": segment_size, "loss": loss,
                "align_loss": align_loss, "duration_loss": duration_loss.detach(),
                "pitch_loss": pitch_loss.detach(), "energy_loss": energy_loss.detach(),
                "_aux": {"log_p_attn": log_p_attn, "durations": durations, "path": path, "p_avg": p_avg,
                         "e_avg": e_avg, "duration_hat": duration_hat, "pitch_hat": pitch_hat,
                         "energy_hat": energy_hat, "decoder_out": y, "segment": segment,
                         "bin_loss": bin_loss.detach(), "forwardsum_loss": forwardsum_loss.detach()}}

    # ------------------------------------------------------------------------------------------ captured decode
    def _graphed_decode(self, h, durations, x_lengths, y_lengths, y_max, am_t0, dev):
        """Upsampler + decoder, then the vocoder, replayed from hipGraphs captured per (B, T_text, y_max, precision).  Returns
        (wav, acoustic-model ms, vocoder ms) with the reference's two timing points."""
        key = (tuple(h.shape), int(y_max), precision.signature())
        ent = self._decode_graphs.get(key)
        if ent is None:
            if len(self._decode_graphs) >= 8:
                self._decode_graphs.pop(next(iter(self._decode_graphs)))
            st = {"h": h.clone(), "d": durations.clone(), "xl": x_lengths.clone(), "yl": y_lengths.clone()}

            def am():
                mask = ~sequence_mask(st["yl"], y_max)
                return self.decoder(self.feature_upsampler(st["h"], st["d"], st["xl"], st["yl"], y_max), mask), mask

            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):                                  # warm-up: allocator pools, weight packs, kernel attributes
                    y, mask = am()
                    self.vocoder(y, f0=None, padding_mask=mask)
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            g_am, g_voc = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            from ..graphs import no_gc_during_capture        # a cyclic-GC pass inside a capture can abort the process (see there)
            with no_gc_during_capture(), torch.cuda.graph(g_am, capture_error_mode="thread_local"):
                st["y"], st["mask"] = am()
            with no_gc_during_capture(), torch.cuda.graph(g_voc, pool=g_am.pool(), capture_error_mode="thread_local"):
                st["wav"] = self.vocoder(st["y"], f0=None, padding_mask=st["mask"])
            ent = self._decode_graphs[key] = (st, g_am, g_voc)
        st, g_am, g_voc = ent
        st["h"].copy_(h); st["d"].copy_(durations); st["xl"].copy_(x_lengths); st["yl"].copy_(y_lengths)
        g_am.replay()
        torch.cuda.synchronize(dev)
        am_infer = (perf_counter() - am_t0) * 1000
        v_t0 = perf_counter()
        g_voc.replay()
        torch.cuda.synchronize(dev)
        return st["wav"], am_infer, (perf_counter() - v_t0) * 1000

    # ------------------------------------------------------------------------------------------ inference
    @torch.inference_mode()
    def synthesise(self, x, x_lengths, sids=None, lids=None, d_factor=1.0, p_factor=1.0, e_factor=1.0,
                   durations_override=None):
        """generator/__init__.py:194-301.  ``durations_override`` is a benchmarking hook (random-init weights
        predict degenerate durations)."""
        dev = x.device
        torch.cuda.synchronize(dev)
        am_t0 = perf_counter()
        x_lengths = x_lengths.to(dev).contiguous()
        Tt = x.shape[1]
        if self.graph_decode and self.graph_encode and x.is_cuda:          # the captured encode builds its own mask / precision scope
            return self._synthesise_body(x, x_lengths, sids, lids, d_factor, p_factor, e_factor, durations_override, dev, am_t0, None)
        input_padding_mask = padding_mask(x_lengths, Tt)
        with precision.index_path():        # durations are integers: their inputs stay exact-f32 in every mode
            return self._synthesise_body(x, x_lengths, sids, lids, d_factor, p_factor, e_factor, durations_override, dev, am_t0,
                                         input_padding_mask)

    def _encode(self, x, x_lengths, input_padding_mask, sids, lids, d_factor, p_factor, e_factor, durations_override):
        """generator/__init__.py:229-258: text encoder, speaker / language embeddings, the three predictors and the frame counts.
        No host reads; called inside ``precision.index_path()`` (the durations' inputs stay exact-f32 in every mode)."""
        dev = x.device
        h, _ = self.text_embedding(x)                                       # :229
        h = self.encoder(h, input_padding_mask)                             # :232
        if (self.num_speakers > 1) and sids is None:
            sids = torch.zeros(x.shape[0], dtype=torch.long, device=dev)
        if (self.num_languages > 1) and lids is None:
            lids = torch.zeros(x.shape[0], dtype=torch.long, device=dev)
        if sids is not None:
            h = h + self.sid_embed(sids.view(-1)).unsqueeze(1)
        if lids is not None:
            h = h + self.lid_embed(lids.view(-1)).unsqueeze(1)
        durations = self.duration_predictor.infer(h, input_padding_mask, factor=d_factor)   # :249
        precision.leave_index_path()        # everything below is continuous: back to the configured precision
        if durations_override is not None:
            durations = durations_override.to(dev).masked_fill(input_padding_mask, 0)
        h, pitch = self.pitch_predictor.infer(h, input_padding_mask, p_factor)              # :252
        h, energy = self.energy_predictor.infer(h, input_padding_mask, e_factor)            # :254
        y_lengths = durations.sum(dim=1)                                                    # :258
        return h, durations, pitch, energy, y_lengths

    def _graphed_encode(self, x, x_lengths, sids, lids, d_factor, p_factor, e_factor, durations_override, dev):
        """_encode replayed from a hipGraph captured per (B, T_text, which optional inputs exist, the three factors, precision): the
        ~60 small launches before the length sync cost their GPU time instead of an interpreter round trip each.  Returns _encode's
        tensors (graph-owned buffers, valid until the next replay of the same graph) and (max frame count, total frame count) from ONE
        host read."""
        key = (tuple(x.shape), sids is not None, lids is not None, float(d_factor), float(p_factor), float(e_factor),
               durations_override is not None, precision.signature())
        ent = self._encode_graphs.get(key)
        Tt = x.shape[1]
        if ent is None:
            if len(self._encode_graphs) >= 8:
                self._encode_graphs.pop(next(iter(self._encode_graphs)))
            st = {"x": x.clone(), "xl": x_lengths.clone(), "sids": None if sids is None else sids.to(dev).clone(),
                  "lids": None if lids is None else lids.to(dev).clone(),
                  "do": None if durations_override is None else durations_override.to(dev).clone()}

            def enc():
                mask = padding_mask(st["xl"], Tt)
                with precision.index_path():
                    h, d, p, e, yl = self._encode(st["x"], st["xl"], mask, st["sids"], st["lids"], d_factor, p_factor, e_factor, st["do"])
                return h, d, p, e, yl, torch.stack([yl.max(), d.sum()])

            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(2):                                  # warm-up: allocator pools, weight packs, kernel attributes
                    enc()
            cur.wait_stream(side)
            torch.cuda.synchronize(dev)
            g_enc = torch.cuda.CUDAGraph()
            from ..graphs import no_gc_during_capture
            with no_gc_during_capture(), torch.cuda.graph(g_enc, capture_error_mode="thread_local"):
                st["out"] = enc()
            ent = self._encode_graphs[key] = (st, g_enc)
        st, g_enc = ent
        st["x"].copy_(x); st["xl"].copy_(x_lengths)
        if sids is not None:
            st["sids"].copy_(sids)
        if lids is not None:
            st["lids"].copy_(lids)
        if durations_override is not None:
            st["do"].copy_(durations_override)
        g_enc.replay()
        h, d, p, e, yl, stats = st["out"]
        y_max, total = stats.tolist()                                   # the one length sync
        return h, d, p, e, yl, int(y_max), int(total)

    def _synthesise_body(self, x, x_lengths, sids, lids, d_factor, p_factor, e_factor, durations_override, dev, am_t0,
                         input_padding_mask):
        if input_padding_mask is None:
            h, durations, pitch, energy, y_lengths, y_max_length, total = self._graphed_encode(
                x, x_lengths, sids, lids, d_factor, p_factor, e_factor, durations_override, dev)
        else:
            h, durations, pitch, energy, y_lengths = self._encode(x, x_lengths, input_padding_mask, sids, lids, d_factor, p_factor,
                                                                  e_factor, durations_override)
            y_max_length = int(y_lengths.max())                                             # data-dependent shape: 1 sync
            total = int(durations.sum())
        if total == 0:                                                                      # alignments.py:152-157
            durations = torch.ones_like(durations)
            y_lengths = durations.sum(dim=1)
            y_max_length = int(y_lengths.max())
        # the padded batch has B * y_max rows of which ``total`` are frames: the fused ConvNeXt MLP skips fully masked row blocks and
        # picks its workgroup mix from the count (kernels.live_rows; a hint: any value gives the same output)
        with K.live_rows(x.shape[0] * y_max_length, total or durations.numel()):
            return self._synthesise_decode(x, x_lengths, h, durations, pitch, energy, y_lengths, y_max_length, dev, am_t0)

    def _synthesise_decode(self, x, x_lengths, h, durations, pitch, energy, y_lengths, y_max_length, dev, am_t0):
        if self.graph_decode and x.is_cuda:
            # BASELINE.json configs[4] "hipGraph-captured decode": everything after the one length sync is shape-static in
            # (B, T_text, y_max_length) -- upsampler + decoder and the vocoder replay from two captured graphs (two, because
            # the reference times the acoustic model and the vocoder separately, :270-284)
            wav, am_infer, v_infer = self._graphed_decode(h, durations, x_lengths, y_lengths.contiguous(), y_max_length, am_t0, dev)
            wav_lengths = y_lengths * self.hop_length
        else:
            target_padding_mask = ~sequence_mask(y_lengths, y_max_length)
            y = self.feature_upsampler(h, durations, x_lengths, y_lengths.contiguous(), y_max_length)   # :263-265
            y = self.decoder(y, target_padding_mask)                                            # :268
            torch.cuda.synchronize(dev)
            am_infer = (perf_counter() - am_t0) * 1000
            v_t0 = perf_counter()
            f0_cond, _ = expand_by_duration(pitch.unsqueeze(-1), durations)                     # :273-276 (unused by WaveNeXt)
            wav = self.vocoder(y, f0=f0_cond, padding_mask=target_padding_mask)                 # :277-281
            wav_lengths = y_lengths * self.hop_length                                           # :282
            torch.cuda.synchronize(dev)
            v_infer = (perf_counter() - v_t0) * 1000
        wav_t = wav.shape[-1] / (self.sample_rate * 1e-3)                                   # :285
        am_rtf, v_rtf = am_infer / wav_t, v_infer / wav_t
        return {"wav": wav.detach().cpu(), "wav_lengths": wav_lengths.detach().cpu(),
                "durations": durations.detach().cpu(), "pitch": pitch.detach().cpu(),
                "energy": energy.detach().cpu(), "am_rtf": am_rtf, "v_rtf": v_rtf, "rtf": am_rtf + v_rtf,
                "latency": am_infer + v_infer}
