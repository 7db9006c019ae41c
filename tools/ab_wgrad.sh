for sh in "4 200 336 256 256 3" "4 100 168 256 256 3" "4 50 84 256 256 3" "4 50 84 1024 256 1" "4 50 84 256 1024 1" "4 100 168 128 128 3" "4 100 168 128 512 1" "4 25 42 512 512 3" "4 25 42 512 2048 1" "4 200 336 256 256 1" "4 200 336 64 64 3"; do
for l in 2 3; do for s in 384 512 768 1024 1536 3072; do
echo -n "L$l/$s "; ALDI_WGRAD_SLOTS=$s ALDI_WGRAD_LEAN=$l timeout 60 python tools/conv_micro.py $sh 20 wgrad 2>&1 | grep wgrad; done; done; done
