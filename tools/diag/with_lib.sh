#!/bin/bash
# run a command with another prebuilt library in place of the shipped one (restored afterwards): with_lib.sh <lib.so> <command ...>
V=$1; shift
LIB=self-paced-contrastive-learning_amd/lib/libspcl_hip.so
cp $LIB /tmp/libspcl_prod.so
trap 'cp /tmp/libspcl_prod.so '$LIB EXIT INT TERM
cp $V $LIB
"$@"
