#!/bin/bash
# AddressSanitizer + UBSan run of the host scaffolding and the CPU oracle on golden fixtures (build container, CPU only:
# sanitizers are not available on the GPU pool).  Every output must match the committed fixture and the sanitizers must
# stay silent.
set -e
cd "$(dirname "$0")/.."
mkdir -p oracle/_build
H=prographmsa_amd/host
FL="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off"
gcc $FL -c -o oracle/_build/pgm_oracle_asan.o oracle/pgm_oracle.c
g++ $FL -std=c++17 -pthread -o oracle/_build/pgmsa_oracle_asan $H/alphabet.cpp $H/model_factory.cpp $H/graph.cpp $H/graph_align.cpp \
    $H/phytree.cpp $H/progressive.cpp $H/distance.cpp $H/csprofile.cpp $H/main.cpp oracle/backend_oracle.cpp oracle/_build/pgm_oracle_asan.o -ldl -lm
B=oracle/_build/pgmsa_oracle_asan G=tests/golden
chk() { local want=$1; shift; local got; got=$("$B" "$@" 2> oracle/_build/asan.err | md5sum | cut -d' ' -f1)
        [ "$got" = "$(md5sum < "$want" | cut -d' ' -f1)" ] && [ ! -s oracle/_build/asan.err ] && echo "ok   $*" || { echo "FAIL $*"; cat oracle/_build/asan.err; exit 1; }; }
chk $G/c1.nw_ml.tree -a -m -T -i 0 $G/c1.fa
chk $G/c1.out.fa --fasta --tree $G/c1.tree $G/c1.fa
chk $G/c1.cs.out.fa --fasta --tree $G/c1.tree --cs_profile $G/K50.lib $G/c1.fa
chk $G/cd1.out.fa --codon --fasta -t $G/cd1.tree $G/cd1.fa
chk $G/m1.out.fa --fasta --tree $G/m1.tree $G/m1.fa
chk $G/x1.out.fa --fasta --tree $G/x1.tree $G/x1.fa
rm -f oracle/_build/pgmsa_oracle_asan oracle/_build/pgm_oracle_asan.o oracle/_build/asan.err
echo "sanitizer run clean"
