#!/bin/bash
O=gpurun_out/r04c; mkdir -p $O
nproc; lscpu | grep -i "numa\|model name\|socket" | head
python3 tools/host_enqueue_probe.py > $O/probe_part.txt 2>&1; cat $O/probe_part.txt | grep -v amdgpu.ids
FGNN_HIP_LIB=$PWD/fgnn-artifacts_amd/lib/libfgnn_hip_prof.so FGNN_HT_PARTITION=0 python3 tools/host_enqueue_probe.py > $O/probe_nopart.txt 2>&1; grep -v amdgpu.ids $O/probe_nopart.txt
