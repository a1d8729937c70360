run() { echo "== $1 $2"; env $2 VQE_HIP_LIB=$PWD/$1 python bench.py --only trainable8 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d.get('kernel_ms'), d.get('value'), d.get('workgroups_per_cu'))"; }
run tools/libvqe_hip_n8_base.so X=1
run tools/libvqe_hip_n8_tile.so X=1
run tools/libvqe_hip_n8_p16.so X=1
run tools/libvqe_hip_n8_tile.so X=1
