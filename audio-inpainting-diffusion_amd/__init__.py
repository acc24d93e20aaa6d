"""MI355X-native EDM inpainting sampling hot path (CQT-octave U-Net denoiser under the Heun/EDM loop).

Drop-in plugin surface for the reference's dotted ``callable`` strings:

    network.callable=audio-inpainting-diffusion_amd.network.Unet_CQT_oct_with_attention
    tester.sampler_callable=audio-inpainting-diffusion_amd.sampler.Sampler
    diff_params.callable=audio-inpainting-diffusion_amd.edm.EDM        (optional; the reference's own EDM also works)

The directory name carries a hyphen, which ``importlib.import_module`` (what the reference's
``dnnlib.call_func_by_name`` uses, utils/dnnlib/util.py:235-297) accepts; for ``import`` statements use the
alias package ``audio_inpainting_diffusion_amd`` at the repo root, whose ``__path__`` is this directory.

Everything numeric runs in hand-written HIP kernels (csrc/*.hip) behind the C-ABI declared in
include/aid_kernels.h; there is no CPU or eager-PyTorch fallback: without the built ``libaid_hip.so``
every operator raises.
"""
__version__ = "0.1.0"
