# md5 of the gfx950 .text of every product object (or of the objects named): the check that an edit changed no machine code
# usage: bash tools/dbg/text_md5.sh [object.o ...]   (default: the product objects of vorta_amd/csrc)
set -e
cd "$(dirname "$0")/../../vorta_amd/csrc"
LLVM=/opt/rocm/lib/llvm/bin
OBJS="$@"
[ -n "$OBJS" ] || OBJS="api.o attn_fwd.o attn_fwd_fp8.o attn_fwd_mx.o attn_fwd_i8.o fp8_quant.o i8_quant.o coreset.o sta_tables.o router.o qk_norm_rope.o mix.o permute.o"
T=$(mktemp -d)
for o in $OBJS; do
  $LLVM/llvm-readelf -S $o | grep -q hip_fatbin || { echo "(no device code)  $o"; continue; }
  $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin $o $T/x.fatbin
  $LLVM/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/x.fatbin --output=$T/x.co --unbundle
  $LLVM/llvm-objcopy -O binary --only-section=.text $T/x.co $T/x.text
  printf "%s  %s  %d bytes\n" "$(md5sum < $T/x.text | cut -d' ' -f1)" "$o" "$(stat -c %s $T/x.text)"
done
rm -rf $T
