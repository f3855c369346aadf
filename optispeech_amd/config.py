"""Configuration values of the reference's ConvNeXt model (configs/model/optispeech.yaml and the files it
composes) as plain dataclasses, and builders that instantiate the modules through the
same ``functools.partial`` contract Hydra uses upstream."""
from dataclasses import dataclass, field
from functools import partial
from types import SimpleNamespace

import torch


@dataclass
class FeatureExtractorArgs:             # configs/data/feature_extractor/22.05khz.yaml over default.yaml
    sample_rate: int = 22050
    n_feats: int = 100
    n_fft: int = 1024
    hop_length: int = 256
    win_length: int = 1024
    f_min: int = 80
    f_max: int = 8000
    center: bool = True


@dataclass
class ModelConfig:
    dim: int = 256                                        # configs/model/optispeech.yaml:10
    n_vocab: int = 250                                    # text_embedding/default.yaml
    text_dropout: float = 0.1
    max_source_positions: int = 2000
    enc_layers: int = 4                                   # encoder/convnext.yaml
    enc_inter: int = 1024
    enc_drop_path: float = 0.2
    dec_layers: int = 4                                   # decoder/convnext.yaml
    dec_inter: int = 1024
    dec_drop_path: float = 0.2
    backbone: str = "convnext"                            # "transformer" = BASELINE configs[3] (encoder/decoder/transformer.yaml)
    tf_heads: int = 2
    tf_units: int = 1024
    tf_blocks: int = 4
    tf_dropout: float = 0.2
    dur: tuple = (2, 384, 3, 0.1)                         # layers, channels, kernel, dropout
    pitch: tuple = (5, 256, 5, 0.5)
    energy: tuple = (2, 384, 3, 0.5)
    embed_kernel: int = 9
    pitch_embed_dropout: float = 0.2
    energy_embed_dropout: float = 0.5
    voc_dim: int = 384                                    # vocoder/wavenext.yaml
    voc_inter: int = 1152
    voc_layers: int = 8
    voc_drop_path: float = 0.1
    segment_size: int = 64                                # generator/default.yaml:12
    lambda_align: float = 5.0
    lambda_duration: float = 1.0
    lambda_pitch: float = 1.0
    lambda_energy: float = 1.0
    lambda_mrd: float = 1.0                               # discriminator/vocos_disc.yaml
    lambda_mel: float = 45.0
    lambda_mr_stft: float = 2.5
    num_speakers: int = 1                                 # data_args.num_speakers (> 1 enables sid_embed, generator/__init__.py:62-63)
    num_languages: int = 1                                # text_processor.num_languages (> 1 enables lid_embed, :64-65)
    fe: FeatureExtractorArgs = field(default_factory=FeatureExtractorArgs)

    def no_dropout(self):
        """Copy with every stochastic rate set to 0 (parity tests)."""
        import copy
        c = copy.deepcopy(self)
        c.text_dropout = c.enc_drop_path = c.dec_drop_path = c.voc_drop_path = 0.0
        c.pitch_embed_dropout = c.energy_embed_dropout = 0.0
        c.tf_dropout = 0.0                                     # the Transformer backbone's dropout / positional / attention rates
        c.dur, c.pitch, c.energy = c.dur[:3] + (0.0,), c.pitch[:3] + (0.0,), c.energy[:3] + (0.0,)
        return c



def _backbones(c):
    """(encoder, decoder) partials: ConvNeXt (configs[1]), the Transformer variant (configs[4]), the LightSpeech
    separable-conv pair, the Conformer or the LeanSpeech LSTM + ConvGLU blocks."""
    if c.backbone == "transformer":
        from .model.transformer import Transformer
        tf = partial(Transformer, attention_heads=c.tf_heads, linear_units=c.tf_units, num_blocks=c.tf_blocks,
                     dropout_rate=c.tf_dropout, positional_dropout_rate=c.tf_dropout, attention_dropout_rate=c.tf_dropout)
        return tf, tf
    if c.backbone == "lightspeech":
        # configs/model/generator/{encoder,decoder}/lightspeech_transformer.yaml
        from .model.lightspeech import LightSpeechTransformerDecoder, LightSpeechTransformerEncoder
        return (partial(LightSpeechTransformerEncoder, kernel_sizes=[5, 25, 13, 9], activation="relu", dropout=0.2),
                partial(LightSpeechTransformerDecoder, kernel_sizes=[17, 21, 9, 13], activation="relu", dropout=0.2,
                        max_source_positions=2000))
    if c.backbone == "conformer":
        # configs/model/generator/{encoder,decoder}/conformer.yaml
        from .model.conformer import Conformer
        kw = dict(attention_heads=2, linear_units=1024, num_blocks=4, dropout_rate=0.2, positional_dropout_rate=0.2,
                  attention_dropout_rate=0.2)
        return partial(Conformer, cnn_module_kernel=7, **kw), partial(Conformer, cnn_module_kernel=31, **kw)
    if c.backbone == "leanspeech":
        # configs/model/generator/{encoder,decoder}/leanspeech.yaml
        from .model.leanspeech import LeanSpeechBackbone
        ls = partial(LeanSpeechBackbone, kernel_size=9, num_layers=4, drop_path=0.2)
        return ls, ls
    from .model.modules import ConvNeXtBackbone
    return (partial(ConvNeXtBackbone, intermediate_dim=c.enc_inter, num_layers=c.enc_layers, drop_path=c.enc_drop_path),
            partial(ConvNeXtBackbone, intermediate_dim=c.dec_inter, num_layers=c.dec_layers, drop_path=c.dec_drop_path))


def make_generator(c: ModelConfig):
    from .model.generator import OptiSpeechGenerator
    from .model.modules import (ConvNeXtBackbone, DurationPredictor, EnergyPredictor, PitchPredictor, TextEmbedding)
    from .model.vocoder import WaveNeXt
    from . import rng
    rng.reset_streams()                   # dropout / DropPath stream ids depend on the construction order inside this model only

    def pred(cls, spec, **kw):
        return partial(cls, num_layers=spec[0], intermediate_dim=spec[1], kernel_size=spec[2], dropout=spec[3],
                       conv_layer_class=torch.nn.Conv1d, **kw)

    loss_coeffs = SimpleNamespace(lambda_align=c.lambda_align, lambda_duration=c.lambda_duration,
                                  lambda_pitch=c.lambda_pitch, lambda_energy=c.lambda_energy)
    return OptiSpeechGenerator(
        dim=c.dim, segment_size=c.segment_size,
        text_embedding=partial(TextEmbedding, n_vocab=c.n_vocab, dropout=c.text_dropout, padding_idx=0,
                               max_source_positions=c.max_source_positions),
        encoder=_backbones(c)[0],
        duration_predictor=pred(DurationPredictor, c.dur),
        pitch_predictor=pred(PitchPredictor, c.pitch, embed_kernel_size=c.embed_kernel,
                             embed_dropout=c.pitch_embed_dropout),
        energy_predictor=pred(EnergyPredictor, c.energy, embed_kernel_size=c.embed_kernel,
                              embed_dropout=c.energy_embed_dropout),
        decoder=_backbones(c)[1],
        vocoder=partial(WaveNeXt, dim=c.voc_dim, intermediate_dim=c.voc_inter, num_layers=c.voc_layers,
                        drop_path=c.voc_drop_path),
        loss_coeffs=loss_coeffs, feature_extractor=c.fe, num_speakers=c.num_speakers, num_languages=c.num_languages,
        data_statistics=None)


def make_optispeech(c: ModelConfig = None, batch_size=32, pretraining_steps=1000, optimizer=None, scheduler=None):
    """Instantiate the full ``OptiSpeech`` module (generator + VocosDiscriminator) the way configs/model/optispeech.yaml
    composes it."""
    from .model.discriminator import VocosDiscriminator
    from .model.generator import OptiSpeechGenerator
    from .model.modules import (ConvNeXtBackbone, DurationPredictor, EnergyPredictor, PitchPredictor, TextEmbedding)
    from .model.optispeech import OptiSpeech, default_args
    from .model.vocoder import WaveNeXt
    from . import rng
    c = c or ModelConfig()
    rng.reset_streams()                   # dropout-site stream ids depend on the construction order inside this model only

    def pred(cls, spec, **kw):
        return partial(cls, num_layers=spec[0], intermediate_dim=spec[1], kernel_size=spec[2], dropout=spec[3],
                       conv_layer_class=torch.nn.Conv1d, **kw)

    gen = partial(
        OptiSpeechGenerator, segment_size=c.segment_size,
        text_embedding=partial(TextEmbedding, n_vocab=c.n_vocab, dropout=c.text_dropout, padding_idx=0,
                               max_source_positions=c.max_source_positions),
        encoder=_backbones(c)[0],
        duration_predictor=pred(DurationPredictor, c.dur),
        pitch_predictor=pred(PitchPredictor, c.pitch, embed_kernel_size=c.embed_kernel,
                             embed_dropout=c.pitch_embed_dropout),
        energy_predictor=pred(EnergyPredictor, c.energy, embed_kernel_size=c.embed_kernel,
                              embed_dropout=c.energy_embed_dropout),
        decoder=_backbones(c)[1],
        loss_coeffs=SimpleNamespace(lambda_align=c.lambda_align, lambda_duration=c.lambda_duration,
                                    lambda_pitch=c.lambda_pitch, lambda_energy=c.lambda_energy))
    voc = partial(WaveNeXt, dim=c.voc_dim, intermediate_dim=c.voc_inter, num_layers=c.voc_layers,
                  drop_path=c.voc_drop_path)
    disc = partial(VocosDiscriminator, loss_coeffs=SimpleNamespace(lambda_mrd=c.lambda_mrd, lambda_mel=c.lambda_mel,
                                                                   lambda_mr_stft=c.lambda_mr_stft))
    train_args, data_args, inference_args = default_args(batch_size, c.fe)
    train_args.pretraining_steps = pretraining_steps
    data_args.num_speakers = c.num_speakers
    if c.num_languages > 1:
        data_args.text_processor.num_languages = c.num_languages
        data_args.text_processor.is_multi_language = True
        data_args.text_processor.languages = [f"lang{i}" for i in range(c.num_languages)]
    return OptiSpeech(dim=c.dim, generator=gen, vocoder=voc, discriminator=disc, train_args=train_args,
                      data_args=data_args, inference_args=inference_args, optimizer=optimizer, scheduler=scheduler)


def synthetic_batch(B=32, T_text=128, T_mel=800, c: ModelConfig = None, seed=1234, ragged=False, device="cpu"):
    """BASELINE.md section 3 synthetic LJSpeech-shaped batch (batch dict schema of TextWavBatchCollate)."""
    c = c or ModelConfig()
    g = torch.Generator().manual_seed(seed)
    if ragged:
        x_len = torch.randint(int(T_text * 0.75), T_text + 1, (B,), generator=g)
        m_len = torch.randint(int(T_mel * 0.75), T_mel + 1, (B,), generator=g)
        x_len[0], m_len[0] = T_text, T_mel
    else:
        x_len = torch.full((B,), T_text, dtype=torch.int64)
        m_len = torch.full((B,), T_mel, dtype=torch.int64)
    x = torch.randint(1, 159, (B, T_text), generator=g)
    x = x * (torch.arange(T_text)[None] < x_len[:, None])
    mvalid = (torch.arange(T_mel)[None] < m_len[:, None])
    mel = torch.randn(B, c.fe.n_feats, T_mel, generator=g) * mvalid[:, None, :]
    pit = torch.randn(B, T_mel, generator=g) * mvalid
    ene = torch.randn(B, T_mel, generator=g) * mvalid
    wav = (torch.rand(B, T_mel * c.fe.hop_length, generator=g) * 2 - 1).clamp_(-1, 1)
    batch = dict(x=x, x_lengths=x_len, mel=mel, mel_lengths=m_len, pitches=pit, energies=ene, wav=wav, sids=None,
                 lids=None, wav_lengths=m_len * c.fe.hop_length, x_texts=[""] * B, filepaths=[""] * B)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()}
