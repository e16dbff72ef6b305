set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
python -c "import phoonnx_amd._ffi as f; f.load(); print('lib ok')" 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "single_plane or reduced_precision or f16_mode" 2>&1 | tail -15 > $O/t1.log
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "f16_vocoder or config4 or config3" 2>&1 | tail -40 > $O/t2.log
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --gen-precision f16 > $O/b_high_f16.json 2> $O/b_high_f16.err
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras > $O/b_high.json 2> $O/b_high.err
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --preset medium --speakers 4 --batch 64 --mixed-lengths --gen-precision f16 --parts 3 > $O/b_c4_f16.json 2> $O/b_c4_f16.err
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --preset medium --speakers 4 --batch 64 --mixed-lengths --parts 3 > $O/b_c4.json 2> $O/b_c4.err
tail -3 $O/t1.log $O/t2.log
