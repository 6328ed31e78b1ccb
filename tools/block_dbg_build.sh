# Experiment builds of the library with the fused draft attention block cut short after a phase (csrc/draft_block.hip, DB_STOP):
#   bash tools/block_dbg_build.sh        -> sequoia_amd/lib/libsequoia_hip_dbstop{1,2,3,4}.so   (git-ignored; they travel with gpurun)
#   SEQUOIA_LIB=$PWD/sequoia_amd/lib/libsequoia_hip_dbstop2.so python tools/block_bench.py 34
cd "$(dirname "$0")/.."
L=sequoia_amd/lib
python -m sequoia_amd.build > /dev/null
for n in 1 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -DDB_STOP=$n -c sequoia_amd/csrc/draft_block.hip -o /tmp/draft_block_dbstop$n.o
  objs=$(ls $L/*.o | grep -v draft_block.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libsequoia_hip_dbstop$n.so $objs /tmp/draft_block_dbstop$n.o
done
ls -la $L/*.so
