# usage (on the GPU box): bash tools/profile_round.sh TAG   -> gpurun_out/prof_TAG/{stats,fetch,write}
TAG=$1; R=/root/repo; O=$R/gpurun_out/prof_$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o s -- python $R/bench.py --steps 10 --warmup 3 --no-predict --no-cpu-baseline --no-graph --no-kernel-events > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f -- python $R/bench.py --steps 3 --warmup 2 --no-predict --no-cpu-baseline --no-graph --no-kernel-events > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/write -o w -- python $R/bench.py --steps 3 --warmup 2 --no-predict --no-cpu-baseline --no-graph --no-kernel-events > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/predict -o p -- python $R/bench.py --predict-only > /dev/null 2>&1
find $O -name "*.db" | head
