#!/usr/bin/env bash
# oracle/ref/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
#
# Builds the executable oracle: the reference's own C++ count+graph path ("path B":
# lib/assembly/src/paths/long/BuildReadQGraph48.cc + HBVFromEdges.cc + MapReduceEngine.h and
# their link closure) compiled FROM THE SOURCES WHERE THEY LIE under /root/reference, with plain
# g++ (the reference's own build system is not run).  Output: oracle/_ref/snref_driver only
# (git-ignored; it travels to the GPU box with the snapshot, the sources never do).
#
# Compiler-compatibility note (stated in DESIGN.md too): four reference files do not compile
# with g++ >= 7 as written.  The recipe applies these 5 one-token edits *in flight* into a
# throw-away overlay (a symlink farm onto the reference tree) under $TMPDIR (never into the repo, never into /root/reference):
#   graph/Digraph.h:1495,1506   From(v).isize( )          -> (int)From(v).size( )
#   kmers/KmerShape.h:566       return getStringId();     -> return KmerShapeId(getStringId());
#   feudal/PQVec.h:53           private:                  -> public:     (Block used by PQVecA)
#   system/System.cc:1134       return ifs;               -> return (bool)(ifs);
# Nothing else of the reference is altered; no header, library or generated file is stood in for.
set -euo pipefail
REF=${SNK_REFERENCE:-/root/reference}
S=$REF/lib/assembly/src
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/../_ref
if [ ! -d "$S" ]; then
  echo "build_ref: $S not present (GPU box?) -- keeping prebuilt oracle/_ref as is"; exit 0
fi
mkdir -p "$OUT"
LIBSNK=$HERE/../../supernova_amd/libsnk.so
if [ -x "$OUT/snref_driver" ] && [ "$OUT/snref_driver" -nt "$HERE/ref_driver.cc" ] && [ "$OUT/snref_driver" -nt "$0" ] &&
   { [ ! -f "$LIBSNK" ] || { [ -x "$OUT/snref_seam" ] && [ "$OUT/snref_seam" -nt "$HERE/seam_driver.cc" ]; }; }; then
  echo "build_ref: oracle/_ref/snref_driver up to date"; exit 0
fi
W=${SNK_REF_WORK:-${TMPDIR:-/tmp}/snk_refbuild.$$}
OV=$W/overlay
mkdir -p "$W/obj"
[ -n "${SNK_REF_WORK:-}" ] || trap 'rm -rf "$W"' EXIT
rm -rf "$OV"
# symlink farm onto the reference tree (quoted #includes resolve relative to the including file,
# so the four patched files must shadow the originals at the same relative path)
cp -rs "$S" "$OV"
rm -f "$OV/graph/Digraph.h" "$OV/kmers/KmerShape.h" "$OV/feudal/PQVec.h" "$OV/system/System.cc"
sed '1495s/From(v).isize( )/(int)From(v).size( )/;1506s/From(v).isize( )/(int)From(v).size( )/' "$S/graph/Digraph.h" > "$OV/graph/Digraph.h"
sed '566s/return getStringId();/return KmerShapeId(getStringId());/' "$S/kmers/KmerShape.h" > "$OV/kmers/KmerShape.h"
sed '53s/private:/public:/' "$S/feudal/PQVec.h" > "$OV/feudal/PQVec.h"
sed '1134s/return \(.*\);/return (bool)(\1);/' "$S/system/System.cc" > "$OV/system/System.cc"

CXX=${CXX:-g++}
OPT=${SNK_REF_OPT:--O3}
# -ffunction-sections / --gc-sections: MarkDups (10X/SecretOps.cc, f4) logs through StatLogger (10X/DfTools.cc); the other
# functions of those two files reach into parts of the reference that do not compile here (paths/long/large/GapToyTools4.cc),
# but nothing the driver calls does -- the linker drops the unreachable functions and with them their open references
FLAGS="-std=c++11 -fpermissive -fopenmp -fno-strict-aliasing -w $OPT -DNDEBUG -ffunction-sections -fdata-sections -I$OV"

# link closure of BuildReadQGraph48.o, paths relative to $S, without .cc
CLOSURE="
10X/MakeHist 10X/Martian 10X/SecretOps 10X/DfTools
Basevector Charvector CompressedSequence Equiv FastIfstream FastaFileset FastaFilestream
FastaConverter FastaFilestreamPreview FastaNameParser FastaVerifier TokenizeString Fastavector Intvector Qualvector Vec VecString
dna/Bases
feudal/BaseVec feudal/BinaryStream feudal/CharString feudal/FeudalControlBlock feudal/FeudalFileReader
feudal/FeudalFileWriter feudal/FieldVec feudal/Generic feudal/Mempool feudal/Oob feudal/PQVec
graph/Digraph kmers/KMerContext kmers/ReadPather
math/Matrix math/Permutation math/PowerOf2
paths/HyperBasevector paths/KmerBaseBroker paths/KmerPath paths/KmerPathInterval
paths/long/ExtendReadPath paths/long/HBVFromEdges paths/long/ReadPath paths/long/ReadPathTools paths/long/ShortKmerReadPather 10X/paths/ReadPathVecX 10X/paths/ReadPathParser 10X/paths/ReadPathX
random/RNGen
system/Assert system/ErrNo system/Exit system/MemTracker system/ProcBuf system/RunTime system/SysConf
system/System system/Thread system/ThreadsafeIO system/UseGDB system/WorklistUtils
system/file/Directory system/file/File system/file/FileReader system/file/FileWriter system/file/SymLink
system/file/TempFile
LinkTimestamp
"
EXTRA=${SNK_REF_EXTRA:-}
compile_one() {
  local rel=$1 src="$OV/$1.cc"
  local obj="$W/obj/$(echo "$rel" | tr / _).o"
  [ -f "$obj" ] && return 0
  $CXX $FLAGS -c "$src" -o "$obj" || { echo "FAILED $rel" >&2; return 1; }
}
export -f compile_one
export CXX FLAGS OV S W
echo "$CLOSURE $EXTRA" | tr ' ' '\n' | grep -v '^$' | xargs -P "$(nproc)" -I{} bash -c 'compile_one {}'
$CXX $FLAGS -c "$HERE/ref_driver.cc" -o "$W/obj/ref_driver.o"
$CXX $FLAGS -DSNK_REF_K60 -c "$HERE/ref_driver.cc" -o "$W/obj/ref_driver60.o"
# archive, so that only the members the driver really reaches are linked (the closure list is a superset)
ar rcs "$W/libref.a" $(ls "$W"/obj/*.o | grep -v -e ref_driver.o -e ref_driver60.o -e LinkTimestamp.o)
$CXX -fopenmp -Wl,--gc-sections -o "$OUT/snref_driver" "$W/obj/ref_driver.o" "$W/obj/LinkTimestamp.o" "$W/libref.a" -lz -lpthread
$CXX -fopenmp -Wl,--gc-sections -o "$OUT/snref_driver60" "$W/obj/ref_driver60.o" "$W/obj/LinkTimestamp.o" "$W/libref.a" -lz -lpthread
# SURVEY 8 row b4 as code: the StageBuildGraph binding of INTEGRATION.md section 1 (seam_driver.cc) compiled against the reference's
# headers and linked with the reference's objects + libsnk.so + the HIP runtime; it needs a GPU to RUN (tests/test_gpu_seam.py)
ROCM=${ROCM_PATH:-/opt/rocm}
if [ -f "$LIBSNK" ] && [ -d "$ROCM/include/hip" ]; then
  $CXX $FLAGS -D__HIP_PLATFORM_AMD__ -I"$ROCM/include" -c "$HERE/seam_driver.cc" -o "$W/obj/seam_driver.o"
  $CXX -fopenmp -Wl,--gc-sections -o "$OUT/snref_seam" "$W/obj/seam_driver.o" "$W/obj/LinkTimestamp.o" "$W/libref.a" -lz -lpthread \
       -L"$(dirname "$LIBSNK")" -lsnk -L"$ROCM/lib" -lamdhip64 -Wl,-rpath,'$ORIGIN/../../supernova_amd' -Wl,-rpath,"$ROCM/lib"
  echo "build_ref: built $OUT/snref_seam (the DF seam stub against the reference's headers)"
fi
echo "build_ref: built $OUT/snref_driver and $OUT/snref_driver60"
