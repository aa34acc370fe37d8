# R5e: conv_pipe's own stamps on the same layers (MPU_PIPE_DEBUG=32: deltas between its stamps; the last delta = partial stores drained)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r5e; mkdir -p $O
cd $R
MPU_CONV_DEEP=0 MPU_PIPE_DEBUG=32 python tools/round5/deep_layers.py 2>&1 | grep -E "pipe stamps|us per launch" | awk '{ if ($1=="pipe") { n=NF; printf "%s %s %s %s %s  first4:", $3,$4,$5,$6,$7; for(i=8;i<12;i++) printf " %s",$i; printf "  ... last3:"; for(i=n-2;i<=n;i++) printf " %s",$i; printf " (n=%d)\n", n-7 } else print }' | tail -60 | tee $O/pipe_stamps.txt
