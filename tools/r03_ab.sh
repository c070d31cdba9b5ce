# A/B of library variants under exp/: kernel times of the bench step for each (same box, same process order)
#   tools/r03_ab.sh tag lib1 lib2 ...
set -e
cd $GRAFT_REPO_ROOT
tag=$1; shift
for lib in "$@"; do
  name=$(basename $lib .so)
  KT_ROWS=${KT_ROWS:-6} bash tools/kernel_times.sh ${tag}_$name $GRAFT_REPO_ROOT/$lib > gpurun_out/${tag}_$name.txt 2>&1
  cat gpurun_out/${tag}_$name.txt
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/kt_${tag}_$name.log | head -1
done
