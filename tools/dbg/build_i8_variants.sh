# variant builds of the int8-score kernel only (the other objects are the product's): libvorta_hip_<name>.so
# usage: bash tools/dbg/build_i8_variants.sh name1 "-DFLAG=.." name2 "-D.." ...
set -e
cd "$(dirname "$0")/../../vorta_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -enable-post-misched=0 -I../../include -I. -Wno-unused-result"
OTHERS="attn_fwd.o attn_fwd_fp8.o attn_fwd_mx.o fp8_quant.o i8_quant.o coreset.o sta_tables.o router.o qk_norm_rope.o mix.o permute.o"
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $f -c attn_fwd_i8.hip -o attn_fwd_i8_v$n.o
  /opt/rocm/bin/hipcc $FLAGS $f -c api.hip -o api_v$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvorta_hip_$n.so attn_fwd_i8_v$n.o api_v$n.o $OTHERS
  echo built libvorta_hip_$n.so "$f"
done
