timeout 2000 python -m pytest tests -m gpu -q -x -rf 2>&1 | grep -v "^RCCL\|^HIP \|^ROCm\|^Hostname\|^Librccl" | tail -60
