mkdir -p gpurun_out/r3b && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q --capture=sys -k "split_fp16_head" > gpurun_out/r3b/tests_mixed.log 2>&1; echo "rc=$?" >> gpurun_out/r3b/tests_mixed.log
grep -E "QSTAT|f32\+h16x2\]|passed|failed|Error|error" gpurun_out/r3b/tests_mixed.log | tail -40
for mv in 0 1; do for rep in 1 2; do echo "mixed_variant $mv: $(CED_MIXED_VARIANT=$mv PRECISION=f32+h16x2 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep Gsamples | tail -1)"; done; done 2>&1 | tee gpurun_out/r3b/field_mixed.txt
echo "f32: $(timeout -k 10 200 python tools/bench_field.py 2>&1 | grep Gsamples | tail -1)" | tee -a gpurun_out/r3b/field_mixed.txt
echo "f16x2: $(PRECISION=f16x2 timeout -k 10 200 python tools/bench_field.py 2>&1 | grep Gsamples | tail -1)" | tee -a gpurun_out/r3b/field_mixed.txt
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r3b/bench_default.json 2> gpurun_out/r3b/bench_default.err; echo "bench rc=$?"
python tools/print_bench_line.py < gpurun_out/r3b/bench_default.json; python -c "import json; d=json.loads([l for l in open(\"gpurun_out/r3b/bench_default.json\") if l.startswith(\"{\")][0]); print({k: d.get(k) for k in (\"value\",\"single_frame_latency_ms\",\"parity_vs_oracle\",\"windows\")}); print(d[\"roofline\"]); print({k:(v.get(\"value\"), v.get(\"parity_vs_oracle\")) for k,v in d.get(\"other_mlp_precisions\",{}).items()})"; tail -5 gpurun_out/r3b/bench_default.err
