export TMPDIR=/tmp
O=gpurun_out/r6_b8; mkdir -p $O
for b in 4 8 16; do INSITU_B=$b INSITU_ROWS=200 timeout 400 python tools/insitu.py > $O/insitu_b$b.txt 2> $O/insitu_b$b.err; head -1 $O/insitu_b$b.txt; done
