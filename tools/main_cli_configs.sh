# The reference's command line on the library (htm-hashjoin_amd/bin/main), BASELINE configs 1-3 and the other operators at
# configs[1]'s size: one JSON line per run, the reference's fields first and in its order. -> gpurun_out/r03_main_cli.txt
M=$GRAFT_REPO_ROOT/htm-hashjoin_amd/bin/main
O=$GRAFT_REPO_ROOT/gpurun_out/r03_main_cli.txt
: > $O
run() { echo "\$ main $*" >> $O; timeout -k 10 600 $M "$@" >> $O 2>&1; echo >> $O; }
run --algo nocc --rSize 1048576 --dataDistr uniform                                   # config 1 (the CPU restatement of the reference's loops)
run --algo atomic --rSize 1048576 --dataDistr uniform                                 # the same on the GPU
run --algo atomic --rSize 134217728 --dataDistr uniform --repeat 5                    # config 2
run --algo atomic --rSize 134217728 --dataDistr local_shuffle --shuffleRange 16 --repeat 5
run --algo htm --rSize 134217728 --dataDistr local_shuffle --shuffleRange 16 --repeat 5
run --algo htm --rSize 134217728 --dataDistr uniform --repeat 5
run --algo auto --rSize 134217728 --dataDistr local_shuffle --shuffleRange 4096 --repeat 5
run --algo prj --rSize 1073741824 --dataDistr local_shuffle --shuffleRange 1024 --repeat 3     # config 3
run --algo atomic --rSize 1073741824 --dataDistr uniform --repeat 3                   # the metric's workload through the CLI
run --algo atomic --rSize 134217728 --dataDistr uniform --gpus 1 --split low          # the sharded library at world 1
cat $O
