# A/B of compile-time flags of hj_build_own.hip on one box: FLAGS="-DX=0|-DX=1" bash tools/dbg/ab_flags.sh
cd $GRAFT_REPO_ROOT
run() { for d in ${DISTS:-uniform:16 sorted:16 local_shuffle:1024}; do set -- ${d%%:*} ${d##*:}; python bench.py --log2n ${LOG2N:-27} --steps 20 --warmup 3 --no-extra --no-cpu-baseline --dist $1 --shuffle-range $2 --build-variant 2 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$TAG | $1', round(d['ms_per_step'],4), round(d['roofline']['launch_us']), d['result']['buildDeferred'], d['result']['conflicts'], d['result']['totalMatches'])"; done; }
IFS='|'; for F in $FLAGS; do unset IFS
touch htm-hashjoin_amd/csrc/hj_build_own.hip; make -C htm-hashjoin_amd/csrc EXTRA_HIPFLAGS="$F" > /dev/null 2>&1
TAG="$F" run
done
touch htm-hashjoin_amd/csrc/hj_build_own.hip; make -C htm-hashjoin_amd/csrc > /dev/null 2>&1
