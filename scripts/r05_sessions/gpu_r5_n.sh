#!/bin/bash
# Round 5, session n: the model-step-in-a-TEXT-block fixture on the device; the 8 MiB stream against the reference binary's file (tests/golden/dropin_rich_8192k.npz)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5n; mkdir -p $O
( timeout 600 python -m pytest tests/test_zgpu_p8stage.py -q -x -p no:cacheprovider -k "media_in_text or mixed_media or wav8m or jpeg_5k or text_32k" 2>&1 | tail -3 ) | tee $O/p8_media_in_text.txt
( timeout 900 python scripts/gpu_long_run.py --bytes 8388608 --out $O/long_run_8m.json 2>&1 | grep -v amdgpu.ids | tail -12 ) | tee $O/long_run_8m.txt
