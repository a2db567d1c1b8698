# partition_us as a function of radixBits (fan-out per pass = 2^(bits/2)) at 2^30; the join's time is not the point here.
R=$GRAFT_REPO_ROOT
for b in ${BITS:-10 12 14 16}; do
  echo "radixBits $b"
  timeout -k 5 120 $R/htm-hashjoin_amd/bin/main --algo prj --rSize ${PRJ_RSIZE:-1073741824} --dataDistr local_shuffle --shuffleRange 1024 --repeat 2 --radixBits $b \
     | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['totalMatches'], 'part', d['partition_us'], 'join', d['join_us'])"
done 2>&1 | tee $R/gpurun_out/prj_bits.txt
