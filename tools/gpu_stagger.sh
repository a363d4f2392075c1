K=gligen_amd/build/kbench
cat > /tmp/st.shapes <<EOS
gemm 32768 2560 320 1 10
gemm 8192 5120 640 1 10
gemm 2048 10240 1280 1 10
gemm 262144 256 512 0 1
conv 4 256 256 256 0 256 1 0 5
conv 4 512 512 128 0 128 1 0 5
EOS
for st in 0 50 100 150; do echo "== stagger $st"; for f in 4,4,1 2,4,1; do GL_GEMM_STAGGER=$st KB_FORCE=$f $K /tmp/st.shapes 20 | grep "^gemm\|^conv" | cut -c1-110; done; done
