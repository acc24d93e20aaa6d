#!/bin/bash
# ONE command that pins the two external dependencies this repository cannot see in its build image:
#   * cqt_nsgt_pytorch  (the reference's constant-Q transform; call sites networks/unet_cqt_oct_with_projattention_adaLN_2.py:620,743,841,
#                        testing/edm_sampler_inpainting.py:63,123)
#   * torchaudio        (torchaudio.functional.resample behind utils/training_utils.py:140-212 resample_batch)
# The VERSION the reference author ran is on record: /root/reference/notebooks/demo_inpainting_spectrogram.ipynb, cell 4 output,
#   "Downloading cqt-nsgt-pytorch-0.0.8.tar.gz (12 kB)" next to torch 1.13.0+cu116 (so torchaudio 0.13.0 is the matching resampler).
# SURVEY.md section 8c calls the package "un-pinned" -- by that evidence it is pinned to 0.0.8; make_cqt_golden.py refuses any other
# version (unless --any-version) and records version + source sha256 in the dump, test_cqt_conformance.py asserts 0.0.8.
# Run it from the repository root on ANY machine where `pip install cqt_nsgt_pytorch==0.0.8 torchaudio` works (CPU is enough):
#
#     tools/pin_external.sh
#
# It (1) dumps the packages' outputs into tests/golden/cqt_ref_*.npz and tests/golden/resample_ref.npz (data only),
#    (2) runs both conformance tests on the CPU oracle, which report which rule preset of audio_inpainting_diffusion_amd/cqt.py::RULE_PRESETS
#        reproduces the package's frame and FAIL if RULES_DEFAULT is a different one,
#    (3) prints the winning preset.  If it is not `default`: set `RULES_DEFAULT = RULE_PRESETS["<name>"]` in cqt.py (host tables only -- no
#        kernel changes; tests/test_cqt_conformance.py::test_hip_cqt_under_alternative_rules_vs_oracle already runs the HIP kernels under the
#        non-default presets) or select it per run with `network.cqt.rules=<name>`.  Commit the .npz files; on the MI355X box
#        `pytest tests/test_cqt_conformance.py tests/test_resample_conformance.py -m gpu` then holds the HIP kernels to the dumps, and
#        harness.load_checkpoint stops refusing trained checkpoints.
set -u
cd "$(dirname "$0")/.."
rc=0
python tests/golden/make_cqt_golden.py || { echo "pin_external: cqt_nsgt_pytorch is not importable here (pip install cqt_nsgt_pytorch==0.0.8)"; rc=1; }
python tests/golden/make_resample_golden.py || { echo "pin_external: torchaudio is not importable here (pip install torchaudio)"; rc=1; }
python -m pytest tests/test_cqt_conformance.py tests/test_resample_conformance.py -q -m "not gpu" -rs || rc=1
python - <<'PY'
import glob, os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
files = sorted(glob.glob("tests/golden/cqt_ref_*.npz"))
if not files:
    print("pin_external: no cqt_ref_*.npz was written -- the CQT stays UNPINNED")
    raise SystemExit(0)
from test_cqt_conformance import _matching_preset
from audio_inpainting_diffusion_amd.cqt import RULES_DEFAULT, RULE_PRESETS
win = None
for f in files:
    match, report = _matching_preset(np.load(f))
    print(f"pin_external: {os.path.basename(f)} -> preset(s) {match or 'NONE'}  {'' if match else report}")
    if match:
        win = set(match) if win is None else (win & set(match))
if win:
    name = sorted(win)[0]
    same = RULE_PRESETS[name] == RULES_DEFAULT
    print(f"pin_external: the package's frame is preset '{name}'" + (" == cqt.RULES_DEFAULT: nothing to change" if same else
          f": set RULES_DEFAULT = RULE_PRESETS['{name}'] in audio_inpainting_diffusion_amd/cqt.py (or run with network.cqt.rules={name})"))
else:
    print("pin_external: no single preset reproduces every dumped configuration -- extend cqt.CQTRules (the report above says which arrays differ)")
PY
exit $rc
