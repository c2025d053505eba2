set -x
nproc; free -g | head -3; df -h /dev/shm /tmp | cat; ulimit -l; ulimit -n
python -c "import torch;print(torch.cuda.device_count())"
rocm-smi --showmeminfo vram | head -8
python tools/rmat_probe.py papers100M 2>&1 | tee gpurun_out/rmat_probe.txt
