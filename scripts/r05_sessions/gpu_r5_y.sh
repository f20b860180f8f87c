#!/bin/bash
# Round 5, session y (the round's last seconds of GPU time): the engine's final probabilities of block 22 (bytes 1 441 792 .. 1 507 327 of the 8 MiB stream)
# against the reference harness's (tmp_longref/harness.block22.p.f32), no digests, report at the first difference
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5y; mkdir -p $O
timeout 98 python scripts/gpu_stage_hashes.py --head-file tmp_longref/stream_head.bin --vocab-file tmp_longref/vocab.bin --detail-block 22 --detail-dir tmp_longref --lean --out $O/detail22.txt > $O/detail22.log 2>&1
tail -4 $O/detail22.log | cut -c1-1500
