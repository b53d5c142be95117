#!/bin/bash
# Long runs of the product loop (development check, not a benchmark): (1) one GPU, conf/cartpole_openai.yaml for 200 000
# generations with a checkpoint every 1000; (2) two ranks sharing the GPU (gloo control plane, peer-store transport, guarded
# run: agreement every comm_check_period generations), 8192 offspring, episodes of at most 20 steps, 60 000 generations.
# Each must end with exit code 0, the expected number of checkpoints and metrics rows, and no exchange time-out.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
out=gpurun_out/soak.txt; : > $out
work=$(mktemp -d /tmp/ses_soak_XXXX)
cp -r simple-es_amd $work/src && cd $work/src && rm -rf logs
t0=$(date +%s)
python run_es.py --cfg-path conf/cartpole_openai.yaml --generation-num 200000 --save-model-period 1000 --seed 1 > $work/one.log 2>&1
rc=$?
d=$(ls -d logs/*/* | head -1)
echo "one GPU: rc $rc, $(( $(date +%s) - t0 )) s, $(ls $d/saved_models | wc -l) checkpoints, $(grep -c episode $d/metrics.jsonl) metrics rows, last: $(grep 'episode:' $work/one.log | tail -1)" >> $R/$out
rm -rf logs
python - <<'PY'
import yaml
c = yaml.load(open("conf/cartpole_openai.yaml"), Loader=yaml.FullLoader)
c["env"]["max_step"] = 20
yaml.dump(c, open("conf/soak.yaml", "w"))
PY
t0=$(date +%s)
SES_DIST_BACKEND=gloo SES_COMM_P2P=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 \
  run_es.py --cfg-path conf/soak.yaml --generation-num 60000 --offspring-num 8192 --save-model-period 5000 --seed 2 > $work/two.log 2>&1
rc=$?
d=$(ls -d logs/*/* | head -1)
echo "two ranks on one GPU: rc $rc, $(( $(date +%s) - t0 )) s, $(ls $d/saved_models | wc -l) checkpoints, $(grep -c episode $d/metrics.jsonl) metrics rows, rollbacks: $(grep -c rollback_to $d/metrics.jsonl), time-outs reported: $(grep -c 'timed out' $work/two.log), last: $(grep 'episode:' $work/two.log | tail -1)" >> $R/$out
tail -3 $work/two.log | cut -c1-300 >> $R/$out
cat $R/$out
