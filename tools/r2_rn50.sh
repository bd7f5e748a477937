#!/bin/bash
# RN50x4 tower: GPU parity tests + throughput + per-family time
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "resnet" > gpurun_out/r2_pytest_rn.txt 2>&1
tail -5 gpurun_out/r2_pytest_rn.txt
cat > /tmp/rn_time.py <<'PY'
import json, os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from capdec_amd import synth, clip as cclip
sd = synth.hot_clip_resnet_state_dict(44, synth.CLIP_RN50X4)
out = {}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
imgs = synth.synthetic_images(n, seed=1, size=288).cuda()
for prec in ("fp32", "fp16"):
    model, _ = cclip.load(sd, device=0, precision=prec)
    model.encode_image(imgs[:8]); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): model.encode_image(imgs)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 3
    eng = model._engine
    eng.profile_enable(True); eng.profile_reset(); model.encode_image(imgs); torch.cuda.synchronize()
    fam = {k: round(v["ms"], 2) for k, v in eng.profile_get().items() if v["launches"]}
    eng.profile_enable(False)
    out[prec] = {"images": n, "ms": round(dt * 1e3, 2), "images_per_s": round(n / dt, 1), "family_ms": fam}
    del model
print(json.dumps(out))
PY
for pk in 1; do
CAPDEC_RN_PACKED=$pk timeout 600 python /tmp/rn_time.py 128 > gpurun_out/r2_rn50_time_packed$pk.json 2> gpurun_out/r2_rn50_time.err
cat gpurun_out/r2_rn50_time_packed$pk.json; tail -2 gpurun_out/r2_rn50_time.err
done
