# variant builds of ONE kernel source (the other objects are the product's): libvorta_hip_<name>.so
# usage: bash tools/dbg/build_variant.sh attn_fwd_fp8 name1 "-DFLAG=.." name2 "-D.." ...
set -e
cd "$(dirname "$0")/../../vorta_amd/csrc"
SRC=$1; shift
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -mllvm -enable-post-misched=0 -I../../include -I. -Wno-unused-result"
ALL="attn_fwd attn_fwd_fp8 attn_fwd_mx attn_fwd_i8 fp8_quant i8_quant coreset sta_tables router qk_norm_rope mix permute"
OTHERS=""
for o in $ALL; do [ "$o" = "$SRC" ] || OTHERS="$OTHERS $o.o"; done
while [ $# -gt 1 ]; do
  n=$1; f=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $f -c $SRC.hip -o ${SRC}_v$n.o
  /opt/rocm/bin/hipcc $FLAGS $f "-DVORTA_VARIANT_FLAGS=\"$f\"" -c api.hip -o api_v$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvorta_hip_$n.so ${SRC}_v$n.o api_v$n.o $OTHERS
  echo built libvorta_hip_$n.so "$f"
done
