"""CPU oracle for the OptiSpeech ConvNeXt hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU fp32 functional code, numpy and one
small C file) of the algorithms on the reference's training-step / ``synthesise``
path.  Every function cites the reference file:line it follows (paths relative to
the upstream tree, mush42/optispeech @ 2024-12-20).

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- as the checker / the timed CPU baseline.
The product package ``optispeech_amd`` never imports it and has no CPU fallback.

Pinning: the reference has no tests and no golden vectors.  The
oracle is pinned against outputs of the reference code itself, imported in the
build container by ``tools/make_golden.py`` (stubs for lightning/omegaconf/hydra/
numba/torchaudio documented there); the resulting fixtures live in
``tests/golden/`` and ``tests/test_oracle_vs_golden.py`` checks every oracle
function against them.  Two pieces are *parity unpinned* and say so:
``losses.mel_filterbank`` / ``losses.mel_l1_loss`` (torchaudio==2.5.1
MelSpectrogram is not installable here; restated from its documented HTK
definition) and the numba row-0 prefix-sum semantics of MAS (numba absent;
sequential fp32 accumulation assumed, see ``alignment.mas_path``).

Layout convention: activations are channels-last ``(B, T, C)`` everywhere (the
reference flips between (B,C,T) and (B,T,C)); parameters are consumed in the
reference's own state-dict schema and shapes.
"""
