# R6z: pipeline tests + CLI tests with the real-loop choice of the producer stream; e2e leg of the bench twice
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6z; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -q -s -x > $O/pytest.log 2>&1; grep -E "passed|failed|bare step|Error|assert " $O/pytest.log | tail -8
for i in 1 2; do timeout 300 python bench.py --e2e-only --steps 60 > $O/e2e_$i.log 2>&1; tail -1 $O/e2e_$i.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); e=d["train_e2e"]; print(d["headline_slices_per_s"], e["serial_slices_per_s"], e["value"], e["fraction_of_headline"], e["producer_stream_candidates_ms_per_step"])'; done
