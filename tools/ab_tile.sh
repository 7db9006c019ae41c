for t in 0 1 2 3; do echo "TILE=$t"
for sh in "4 200 336 256 256 3" "4 100 168 256 256 3" "4 200 336 64 256 1" "4 200 336 256 64 1" "4 100 168 128 128 3" "4 100 168 512 128 1" "4 50 84 256 1024 1"; do
ALDI_IGEMM_TILE=$t python tools/conv_micro.py $sh 20; done; done
