# pipelined hybrid step with the SPEEDY window and the reservoir readout on disjoint CU sets; run on the GPU box from the repo root
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "sequential" > gpurun_out/cusplit.log
python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/cusplit.log
for C in 0 32 64 96 128 160; do
  echo "pipeline speedy_cus=$C plain readout" >> gpurun_out/cusplit.log
  SML_PIPELINE=1 SML_SPEEDY_CUS=$C SML_PERSISTENT_READOUT=0 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 30 2>>gpurun_out/cusplit.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])" >> gpurun_out/cusplit.log
done
cat gpurun_out/cusplit.log; tail -3 gpurun_out/cusplit.err
