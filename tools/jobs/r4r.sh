mkdir -p gpurun_out/r4r
RCF_H2P=1 python tools/ab_korder.py fp32 > gpurun_out/r4r/h2p1.txt 2>&1
python tools/ab_korder.py fp32 > gpurun_out/r4r/h2pdef.txt 2>&1
for f in h2pdef h2p1; do echo "== $f"; grep -v amdgpu gpurun_out/r4r/$f.txt | cut -c1-40,100-215; done
RCF_H2P=1 python tools/bench_h2p.py > gpurun_out/r4r/bench_h2p.txt 2>&1; cut -c1-260 gpurun_out/r4r/bench_h2p.txt
