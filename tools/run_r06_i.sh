#!/bin/bash
# round 6: latency work on the small kernels (LayerNorm + modulation: batched row requests, modulation rows up front, both streams of a double
# block in one launch; the K-slice reduce pass: all requests up front) -- tests, then A/Bs at batch 1 and at the headline batch
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r06i; mkdir -p $out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py tests/test_model_gpu.py tests/test_pipeline_gpu.py tests/test_configs_gpu.py tests/test_text_encoders_gpu.py -q -m gpu 2>&1 | tail -8 | cut -c1-250
b1() { timeout 400 python bench.py --no-cpu-baseline --no-attention-ab --no-pil-delta --no-peak-probe --batch 1 --steps 4 --warmup 2 "$@" 2>&1 | grep '^{"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['sec_per_img_per_gpu'],4), round(d['dit_algorithmic_tflops_per_gpu']), round(d['roofline']['achieved']), round(d['roofline']['attention']['achieved']))"; }
for r in 1 2; do
  echo "== SL512 b1 round-5 behaviour (ln_joint 0, ln_prefetch 0)"; b1 --height 576 --width 512 --option ln_joint=0 --option ln_prefetch=0
  echo "== SL512 b1 ln_joint 1, ln_prefetch 0"; b1 --height 576 --width 512 --option ln_prefetch=0
  echo "== SL512 b1 default (ln_joint 1, ln_prefetch by size)"; b1 --height 576 --width 512
  echo "== P1024 b1 round-5 behaviour"; b1 --option ln_joint=0 --option ln_prefetch=0
  echo "== P1024 b1 default"; b1
done 2>&1 | tee $out/b1_ab.log
timeout 600 python tools/dit_ab.py ln_prefetch=0,1 2>&1 | grep "ms/forward" | tee $out/ab_b8_ln_prefetch.log
