"""Attribute-dict configuration objects shaped like the reference's hydra config tree.

The reference hands its components the full hydra ``DictConfig`` (utils/setup.py:40-66).  Our classes only
use attribute access (``args.network.cqt.num_octs`` ...), so an omegaconf ``DictConfig``, the reference's
``dnnlib.EasyDict`` or the ``Cfg`` below all work.  ``make_args`` builds the trees for the shipped
configurations that BASELINE.json names, with values copied from the reference YAML files cited inline.
"""
from __future__ import annotations

import copy


class Cfg(dict):
    """dict with attribute access, recursively applied to nested dicts."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        for k, v in list(self.items()):
            if isinstance(v, dict) and not isinstance(v, Cfg):
                self[k] = Cfg(v)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = Cfg(value) if isinstance(value, dict) and not isinstance(value, Cfg) else value

    def __deepcopy__(self, memo):
        return Cfg({k: copy.deepcopy(v, memo) for k, v in self.items()})


_ATTN = dict(num_heads=8, attn_dropout=0.0, bias_qkv=False, N=0, rel_pos_num_buckets=32,
             rel_pos_max_distance=64, use_rel_pos=False, Nproj=8)

# conf/network/paper_1912_unet_cqt_oct_attention_adaLN_2.yaml:6-64
NETWORK_22K = dict(
    name="unet_cqt_oct_with_attention",
    callable="audio_inpainting_diffusion_amd.network.Unet_CQT_oct_with_attention",
    use_fencoding=False, use_norm=True, filter_out_cqt_DC_Nyq=True, depth=7, emb_dim=256,
    Ns=[64, 96, 96, 128, 128, 256, 256], attention_layers=[0, 0, 0, 0, 1, 1, 1, 1], Ss=[2] * 7,
    num_dils=[2, 3, 4, 5, 6, 7, 7], cqt=dict(window="kaiser", beta=1, num_octs=7, bins_per_oct=64),
    bottleneck_type="res_dil_convs", num_bottleneck_layers=1, attention_dict=_ATTN)

# conf/network/paper_1912_unet_cqt_oct_attention_44k_2.yaml
NETWORK_44K = dict(
    name="unet_cqt_oct_with_attention",
    callable="audio_inpainting_diffusion_amd.network.Unet_CQT_oct_with_attention",
    use_fencoding=False, use_norm=True, filter_out_cqt_DC_Nyq=True, depth=8, emb_dim=256,
    Ns=[64, 64, 96, 96, 128, 128, 256, 256], attention_layers=[0, 0, 0, 0, 0, 1, 1, 1, 1], Ss=[2] * 7,
    num_dils=[2, 3, 4, 5, 6, 7, 8, 8], cqt=dict(window="kaiser", beta=1, num_octs=8, bins_per_oct=64),
    bottleneck_type="res_dil_convs", num_bottleneck_layers=1, attention_dict=_ATTN)

# conf/tester/inpainting_tester.yaml:20-57 (T is overridden per BASELINE.json config)
TESTER_LONG = dict(
    name="inpainting_tester", sampler_callable="audio_inpainting_diffusion_amd.sampler.Sampler",
    T=35, order=2, filter_out_cqt_DC_Nyq=True,
    posterior_sampling=dict(xi=0.25, norm=2, smoothl1_beta=1),
    data_consistency=dict(use=True, type="always", smooth=True, hann_size=50),
    diff_params=dict(same_as_training=False, sigma_data=0.063, sigma_min=1e-4, sigma_max=1, P_mean=-1.2,
                     P_std=1.2, ro=13, ro_train=13, Schurn=10, Snoise=1.0, Stmin=0, Stmax=50),
    spectrogram_inpainting=dict(stft=dict(window="hann", n_fft=1024, hop_length=256, win_length=1024),
                                time_mask_length=2000, time_start_idx="None", min_masked_freq=300, max_masked_freq=2000),
    inpainting=dict(mask_mode="long", long=dict(gap_length=1500, start_gap_idx="None"),
                    short=dict(num_gaps=4, gap_length=25, start_gap_idx="None")))

# conf/diff_params/edm.yaml
DIFF_PARAMS = dict(callable="audio_inpainting_diffusion_amd.edm.EDM", sigma_data=0.063, sigma_min=1e-5,
                   sigma_max=10, P_mean=-1.2, P_std=1.2, ro=13, ro_train=10, Schurn=5, Snoise=1, Stmin=0,
                   Stmax=50, aweighting=dict(use_aweighting=False, ntaps=101))


def make_args(name: str = "maestro22k", audio_len: int = 184184, T: int = 36, gap_ms: float = 300.0,
              xi: float = 0.25) -> Cfg:
    """Config trees for BASELINE.json's workloads.

    maestro22k   : exp=maestro22k_8s (fs 22050, L 184184, conf/exp/maestro22k_8s.yaml:51-52) + 7-octave net
    librispeech16k: exp=librispeech16k_8s (fs 16000, same L) + 7-octave net + short-gap tester (T 70, hann 100)
    musicnet44k  : exp=musicnet44k_4s/8s (fs 44100) + 8-octave net
    maestro22k_noattention : same as maestro22k with conf/network/paper_1912_unet_cqt_oct_noattention_adaln.yaml
                   (attention_layers all 0, use_rel_pos: True -- unused without attention blocks)
    """
    if name == "maestro22k_noattention":
        fs, net = 22050, copy.deepcopy(NETWORK_22K)
        net["attention_layers"] = [0] * 8
        net["attention_dict"]["use_rel_pos"] = True
    elif name == "maestro22k":
        fs, net = 22050, NETWORK_22K
    elif name == "librispeech16k":
        fs, net = 16000, NETWORK_22K
    elif name == "musicnet44k":
        fs, net = 44100, NETWORK_44K
    else:
        raise ValueError(name)
    tester = copy.deepcopy(TESTER_LONG)
    tester["T"] = T
    tester["posterior_sampling"]["xi"] = xi
    tester["inpainting"]["long"]["gap_length"] = gap_ms
    if name == "librispeech16k":  # conf/tester/inpainting_tester_shortgaps.yaml:20,37
        tester["data_consistency"]["hann_size"] = 100
        tester["inpainting"]["mask_mode"] = "short"
        tester["inpainting"]["short"]["gap_length"] = gap_ms
    return Cfg(network=copy.deepcopy(net), tester=tester, diff_params=copy.deepcopy(DIFF_PARAMS),
               exp=dict(sample_rate=fs, audio_len=audio_len, exp_name=name))


def small_args(num_octs=4, bins_per_oct=8, Ns=(8, 8, 16, 16), num_dils=(1, 2, 2, 3), attention=(0, 0, 1, 1, 1),
               audio_len=4096, fs=22050, emb_dim=32, T=4, xi=0.0, use_fencoding=False, bias_qkv=False, use_rel_pos=False) -> Cfg:
    """Reduced-size network of the same family, for parity tests that finish in seconds on CPU."""
    a = make_args("maestro22k", audio_len=audio_len, T=T, xi=xi)
    a.network.Ns, a.network.num_dils = list(Ns), list(num_dils)
    a.network.attention_layers = list(attention)
    a.network.emb_dim = emb_dim
    a.network.depth = num_octs
    a.network.cqt.num_octs, a.network.cqt.bins_per_oct = num_octs, bins_per_oct
    a.network.use_fencoding = bool(use_fencoding)
    a.network.attention_dict.bias_qkv, a.network.attention_dict.use_rel_pos = bool(bias_qkv), bool(use_rel_pos)
    a.exp.sample_rate = fs
    return a
