cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/emul512 --output-format csv -- python3 $R/tools/emulate_rank.py --grid 512 --ranks 8 --steps 6 > $R/gpurun_out/emul512.log 2>&1
tail -2 $R/gpurun_out/emul512.log
ls $R/gpurun_out/emul512/*/
