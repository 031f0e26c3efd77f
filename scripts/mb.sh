python scripts/conv_bench.py 128 14 14 256 256 3 1 0 20
python scripts/conv_bench.py 128 14 14 256 256 3 1 3 20
python scripts/conv_bench.py 128 14 14 256 256 3 1 1 20
python scripts/conv_bench.py 334 14 14 256 256 3 1 3 20
python scripts/conv_bench.py 128 28 28 128 128 3 1 0 20
python scripts/conv_bench.py 128 28 28 128 128 3 1 3 20
python scripts/conv_bench.py 128 7 7 512 512 3 1 0 20
python scripts/conv_bench.py 128 7 7 512 512 3 1 3 20
