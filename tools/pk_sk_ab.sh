#!/bin/bash
# same-box A/B of the stream-K (window) launches: var/libdmh_pk0.so (scalar transforms) against the tree's library
export WINO_SHAPES="512,256,10,32,0;512,256,20,28,0;256,128,18,24,0;128,64,50,64,0;256,128,30,40,0;64,64,56,78,1"
for r in 1 2; do
echo "== K10 stream-K shapes, scalar"; DMH_HIP_LIB=var/libdmh_pk0.so python3 tools/wino_bench.py 12 20 | grep custom | cut -c1-40,60-100,130-200
echo "== K10 stream-K shapes, packed"; python3 tools/wino_bench.py 12 20 | grep custom | cut -c1-40,60-100,130-200
done
echo "== K17, scalar"; DMH_HIP_LIB=var/libdmh_pk0.so python3 tools/wino32_bench.py 12 20 nomiopen | cut -c1-60,90-200
echo "== K17, packed"; python3 tools/wino32_bench.py 12 20 nomiopen | cut -c1-60,90-200
echo "== K17, scalar"; DMH_HIP_LIB=var/libdmh_pk0.so python3 tools/wino32_bench.py 12 20 nomiopen | cut -c1-60,90-200
echo "== K17, packed"; python3 tools/wino32_bench.py 12 20 nomiopen | cut -c1-60,90-200
