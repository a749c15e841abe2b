set -e
# var/libwino_ablpk0.so: scalar transforms (DMH_WINO_PK=0, round-5 epilogue); pk1: packed input transform; pk2: + packed epilogue;
# pk4: + the accumulators read once by asm v_accvgpr_read_b32
for sh in "64 64 80 256 1 12" "256 256 20 64 1 12" "128 64 80 256 0 12" "64 64 80 256 1 32" "128 128 40 128 1 32" "32 64 80 256 2 32"; do
  python3 tools/wino_ablate.py $sh pk0 pk2 pk4 pk0 pk2 pk4
done
