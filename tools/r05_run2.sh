cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -12 > $O/pytest_gpu_a.txt
IAF_FUSE_PAIR=1 python tools/fused_stamps.py --hw 8 2>&1 | grep -v amdgpu.ids > $O/stamps2_pair.txt
python bench.py --train --model --steps 20 --warmup 5 > $O/bench_train_model_a.json 2> $O/bench_train_model_a.err
python bench.py --iw-eval --model --steps 50 --warmup 5 > $O/bench_iw_model_a.json 2> $O/bench_iw_model_a.err
python bench.py --no-cpu-baseline > $O/bench2.json 2> /dev/null
cat $O/pytest_gpu_a.txt $O/stamps2_pair.txt; tail -3 $O/bench_train_model_a.err; tail -3 $O/bench_iw_model_a.err
python - <<'PY'
import json
for f in ("bench_train_model_a", "bench_iw_model_a", "bench2"):
    try:
        d = json.load(open("gpurun_out/r05/%s.json" % f))
        print(f, d["ms_per_step"], d["value"], json.dumps(d.get("exchange", {}))[:900], json.dumps({k: v for k, v in d["config"].items() if "ms_" in k or "speedup" in k}))
    except Exception as e:
        print(f, "FAILED", e)
PY
