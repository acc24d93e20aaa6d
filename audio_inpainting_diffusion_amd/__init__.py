"""MI355X-native EDM inpainting sampling hot path (CQT-octave U-Net denoiser under the Heun/EDM loop).

Drop-in plugin surface for the reference's dotted ``callable`` strings:

    network.callable=audio_inpainting_diffusion_amd.network.Unet_CQT_oct_with_attention
    tester.sampler_callable=audio_inpainting_diffusion_amd.sampler.Sampler
    diff_params.callable=audio_inpainting_diffusion_amd.edm.EDM        (optional; the reference's own EDM also works)

``audio-inpainting-diffusion_amd`` (the repository's hyphenated name) is an alias directory that registers these
same module objects under that name, so both spellings resolve to ONE set of classes and one loaded library.

Everything numeric runs in hand-written HIP kernels (csrc/*.hip) behind the C-ABI declared in
include/aid_kernels.h; there is no CPU or eager-PyTorch fallback: without the built ``libaid_hip.so``
every operator raises.
"""
__version__ = "0.2.0"
