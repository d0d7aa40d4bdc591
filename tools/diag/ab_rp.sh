set -e
P=self-paced-contrastive-learning_amd
b() { for i in 1 2; do python bench.py --no-cpu-baseline --no-extras --steps 50 2>/dev/null | grep -o "ms_per_step[^,]*"; done; }
echo "RP narrow 22"; b
sed -i 's/#define SPCL_FAST_RP_NARROW 22/#define SPCL_FAST_RP_NARROW 16/' $P/csrc/conv_fast.hip; python $P/build.py > /dev/null 2>&1
echo "RP narrow 16"; b
sed -i 's/#define SPCL_FAST_RP_NARROW 16/#define SPCL_FAST_RP_NARROW 22/' $P/csrc/conv_fast.hip; touch $P/csrc/conv_fast.hip; python $P/build.py > /dev/null 2>&1
echo "RP narrow 22"; b
