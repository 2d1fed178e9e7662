#!/bin/bash
# build a variant of the library with extra flags for ONE source file (a complete build in a scratch copy of csrc/):
#   tools/build_variant.sh NAME FILE "-DFLAG=1"      e.g.  tools/build_variant.sh c512 mapper "-DGS_MAP_CHUNK=512"
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TMP=$(mktemp -d)
mkdir -p $TMP/pkg/csrc $TMP/include $ROOT/tools/ubench/bin
cp $ROOT/taichi_gaussian_rasterizer_amd/csrc/*.hip $ROOT/taichi_gaussian_rasterizer_amd/csrc/*.cpp \
   $ROOT/taichi_gaussian_rasterizer_amd/csrc/*.h $ROOT/taichi_gaussian_rasterizer_amd/csrc/Makefile $TMP/pkg/csrc/
cp $ROOT/include/*.h $TMP/include/
cd $TMP/pkg/csrc
# the per-file rules of the Makefile carry their own flags: append the variant's to the one file's rule
python3 - "$2" "$3" <<'PY'
import re, sys
name, extra = sys.argv[1], sys.argv[2]
s = open("Makefile").read()
s = s.replace("../../include/", "../../include/")
if f"$(OBJ)/{name}.o:" in s:
    s = re.sub(r"(\$\(OBJ\)/%s\.o:[^\n]*\n\t@mkdir[^\n]*\n\t\$\(HIPCC\) \$\(COMMON\))" % name, r"\1 " + extra, s)
else:
    s = s.replace("$(OBJ)/%.o: %.hip $(HDRS)", "$(OBJ)/%s.o: %s.hip $(HDRS)\n\t@mkdir -p $(OBJ)\n\t$(HIPCC) $(COMMON) %s -c $< -o $@\n\n$(OBJ)/%%.o: %%.hip $(HDRS)" % (name, name, extra))
open("Makefile", "w").write(s)
PY
make -j6 > $TMP/build.log 2>&1 || { tail -20 $TMP/build.log; exit 1; }
cp $TMP/pkg/libgsplat_hip.so $ROOT/tools/ubench/bin/lib_$1.so
rm -rf $TMP
echo built tools/ubench/bin/lib_$1.so
