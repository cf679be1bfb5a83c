for v in ${VARS:-- B N G H}; do echo "variant=$v"; JSG_1024_VARIANT=$v timeout -k 10 200 python tools/kbench.py --set frames 2>&1 | grep -v amdgpu.ids | grep -E '"frames": (4096|16384|262144)'; done
