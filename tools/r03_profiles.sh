# the committed profiles of round 3 (run on the GPU box):  bash tools/r03_profiles.sh <tag>
set -e
cd $GRAFT_REPO_ROOT
tag=${1:-r03_v3}
bash tools/profile_all.sh ${tag} all
bash tools/sq_counters.sh ${tag}
bash tools/profile_all.sh ${tag}_config5 all --config 5 --steps 4 --warmup 2
bash tools/profile_all.sh ${tag}_config6 stats --config 6 --steps 10 --warmup 3
python bench.py --steps 200 --warmup 10 > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
ls gpurun_out | grep ${tag}
