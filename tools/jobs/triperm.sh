set -x
R=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in librtamd.so variants/librtamd_triperm.so; do
    D=/tmp/triperm_$$
    rm -rf $D
    RTAMD_LIB=$R/rust-raytracer_amd/$lib timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d $D -- python3 $R/bench.py --inner --steps 1 --warmup 0 > /dev/null 2> $D.err || exit 1
    echo "== $lib" >> $R/gpurun_out/r3_triperm.txt
    python3 $R/tools/pmc_summary.py "$(find $D -name '*counter_collection.csv' | head -1)" | grep -A8 "k_wave_trace" >> $R/gpurun_out/r3_triperm.txt
    rm -rf $D $D.err
done
cd $R
for rep in 1 2 3; do
for lib in rust-raytracer_amd/librtamd.so rust-raytracer_amd/variants/librtamd_triperm.so; do
  bash tools/sweep_opts.sh gpurun_out/r3_triperm.txt $lib "chunk=64" --steps 20 || exit 1
done
done
python -u -m pytest tests/test_gpu_trace.py tests/test_gpu_render.py -x -q --timeout 300 --timeout-method thread 2>&1 | tail -3 >> gpurun_out/r3_triperm.txt
