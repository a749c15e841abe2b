for sh in "64 64 80 256 1 32" "256 256 20 64 1 12" "128 64 80 256 0 12"; do python3 tools/wino_ablate.py $sh 0 128 384 16 144 400 0; done
