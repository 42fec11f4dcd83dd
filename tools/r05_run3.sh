cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
timeout 300 python -m pytest tests/test_hip_generic_backward.py -q 2>&1 | tail -25 > $O/t_generic.txt
timeout 200 python bench.py --train --model --steps 20 --warmup 5 > $O/bench_train_model_b.json 2> $O/bench_train_model_b.err
IAF_BENCH_SEGMENT_GRAPHS=1 timeout 200 python bench.py --train --model --steps 20 --warmup 5 > $O/bench_train_model_b_seg.json 2> /dev/null
timeout 200 python bench.py --train --layers --steps 20 --warmup 5 > $O/bench_train_layers_b.json 2> $O/bench_train_layers_b.err
timeout 100 python bench.py --train --steps 50 > $O/bench_train_b.json 2> $O/bench_train_b.err
cat $O/t_generic.txt; grep -i "refused\|error\|Traceback" $O/*_b.err | head
python - <<'PY'
import json
for f in ("bench_train_model_b", "bench_train_model_b_seg", "bench_train_layers_b", "bench_train_b"):
    try:
        d = json.load(open("gpurun_out/r05/%s.json" % f)); e = d.get("exchange", {})
        print(f, round(d["ms_per_step"], 3), e.get("launch"), {k: e.get(k) for k in ("buckets", "step_without_exchange_ms", "exposed_ms", "rccl")})
    except Exception as ex:
        print(f, "FAILED", ex)
PY
