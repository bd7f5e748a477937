#!/bin/bash
# RN50x4 tower: GPU parity tests + throughput + per-kernel stats
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_rn.txt 2>&1
tail -5 gpurun_out/r2_pytest_rn.txt
cat > /tmp/rn_time.py <<'PY'
import json, sys, time, torch
sys.path.insert(0, ".")
from capdec_amd import synth, clip as cclip
sd = synth.hot_clip_resnet_state_dict(44, synth.CLIP_RN50X4)
out = {}
for prec in ("fp32", "fp16"):
    model, _ = cclip.load(sd, device=0, precision=prec)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    imgs = synth.synthetic_images(n, seed=1, size=288).cuda()
    model.encode_image(imgs[:8]); torch.cuda.synchronize()
    t0 = time.time(); 
    for _ in range(3): model.encode_image(imgs)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    out[prec] = {"images": n, "ms": dt * 1e3, "images_per_s": n / dt}
    del model
print(json.dumps(out))
PY
timeout 600 python /tmp/rn_time.py 128 > gpurun_out/r2_rn50_time.json 2> gpurun_out/r2_rn50_time.err
cat gpurun_out/r2_rn50_time.json; tail -3 gpurun_out/r2_rn50_time.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rnprof -o rn -- python /tmp/rn_time.py 64 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(find /tmp/rnprof -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && head -25 "$f" > gpurun_out/r2_rn50_kernel_stats.csv
