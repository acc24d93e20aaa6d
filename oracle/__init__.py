"""CPU oracle for the EDM-inpainting hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import anything from this package.  The product package
(``audio_inpainting_diffusion_amd``) never imports it and fails loudly when its HIP
extension is missing.

Pinning status
--------------
* ``unet.py`` / ``edm.py`` / ``sampler.py``: pinned against golden vectors captured by importing
  the reference itself in the dev container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``,
  checked by ``tests/test_oracle_golden.py``).
* ``nsgt_cqt.py``: **parity unpinned**.  The reference delegates its CQT to the un-vendored,
  un-pinned PyPI package ``cqt_nsgt_pytorch`` (reference ``networks/unet_cqt_oct_with_projattention_adaLN_2.py:9,620``)
  which is absent from /root/reference and from this image.  The restatement follows the published
  NSGT algorithm (Holighaus, Doerfler, Velasco, Grill 2013; Balazs et al. 2011 painless dual frame)
  and is validated by mathematical properties only (perfect reconstruction, adjointness, tone
  localisation, shape contract of the call sites).
"""
