"""Pins the CPU oracle (oracle/pgm_oracle.c) + host scaffolding (prographmsa_amd/host) against outputs of the
reference's prebuilt binary committed under tests/golden/ (see tests/golden/make_golden.py).  CPU only."""
import hashlib
import json
import os
import subprocess

import pytest

import gen

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def run_oracle(oracle_build, args, cwd=None):
    exe = os.path.join(oracle_build, "pgmsa_oracle")
    r = subprocess.run([exe] + args, capture_output=True, text=True, cwd=cwd)
    assert r.returncode == 0, r.stderr
    return r.stdout


def gold(name):
    return open(os.path.join(GOLD, name)).read()


@pytest.mark.parametrize("case", ["c1", "c2", "m1", "x1"])
def test_fasta_identical_to_reference(oracle_build, case):
    out = run_oracle(oracle_build, ["--fasta", "--tree", os.path.join(GOLD, case + ".tree"), os.path.join(GOLD, case + ".fa")])
    assert out == gold(case + ".out.fa")


def test_pairwise_alignments_identical(oracle_build, tmp_path):
    pairs = json.load(open(os.path.join(GOLD, "pairs.json")))
    bad = []
    for seed, p in pairs.items():
        (tmp_path / "p.fa").write_text(p["fasta"])
        (tmp_path / "p.tree").write_text(p["tree"])
        if run_oracle(oracle_build, ["--fasta", "--tree", str(tmp_path / "p.tree"), str(tmp_path / "p.fa")]) != p["out"]:
            bad.append(seed)
        if run_oracle(oracle_build, ["--fasta", "-m", "--tree", str(tmp_path / "p.tree"), str(tmp_path / "p.fa")]) != p["out_mldist"]:
            bad.append(seed + "m")
    assert not bad, bad


@pytest.mark.parametrize("batched", [False, True])
def test_nw_distance_pairs(oracle_build, tmp_path, batched, monkeypatch):
    """alignPair + computeDistance: 2-sequence `-a [-m] -T -i 0` prints (b:d/2,a:d/2); at 6 s.f.  batched: the distances come
    from the oracle's pgmo_mldist (oracle/pgm_oracle.c, the checker of the device kernel pgm_mldist_kernel) instead of the
    host mirror's estimator — both follow src/DistanceFactoryML.h:66-190."""
    if batched:
        monkeypatch.setenv("PGM_DEVICE_MLDIST", "1")
    nw = json.load(open(os.path.join(GOLD, "nw_pairs.json")))
    for seed, p in nw.items():
        (tmp_path / "p.fa").write_text(p["fasta"])
        assert run_oracle(oracle_build, ["-a", "-m", "-T", "-i", "0", str(tmp_path / "p.fa")]) == p["ml"], seed
        assert run_oracle(oracle_build, ["-a", "-T", "-i", "0", str(tmp_path / "p.fa")]) == p["pdist"], seed


def test_ancestral_sequences_and_profiles(oracle_build, tmp_path):
    """Merge numerics (mergeGraphs' node profiles, GraphAlign.h:569-620 = pgmo_merge_profiles through the oracle backend):
    --ancestral_seqs --profile_out prints every ancestor's profile at 6 significant digits and adds the ancestral rows to the
    FASTA (ProgressiveAlignment.h:289-411, profile.h).  3 x 30 family in full text, 8 x 120 / 64 x 400 / 6 x 60 codons by md5."""
    md5 = json.load(open(os.path.join(GOLD, "md5.json")))
    prof = str(tmp_path / "p.prof")
    out = run_oracle(oracle_build, ["--fasta", "--ancestral_seqs", "--profile_out", prof, "-t", os.path.join(GOLD, "a1.tree"), os.path.join(GOLD, "a1.fa")])
    assert out == gold("a1.anc.out.fa")
    assert open(prof).read() == gold("a1.anc.prof")
    for name, flags in (("c1", []), ("c2", []), ("cd1", ["--codon"])):
        out = run_oracle(oracle_build, flags + ["--fasta", "--ancestral_seqs", "--profile_out", prof, "-t", os.path.join(GOLD, name + ".tree"), os.path.join(GOLD, name + ".fa")])
        assert hashlib.md5(out.encode()).hexdigest() == md5[name + ".anc.out.fa"], name
        if name + ".anc.prof" in md5:   # (amino acids; the codon profiles are not reproducible to 6 digits: ECM expm vs Eigen's EigenSolver)
            assert hashlib.md5(open(prof, "rb").read()).hexdigest() == md5[name + ".anc.prof"], name


@pytest.mark.parametrize("case,flags", [
    # 8 taxa: the last BioNJ join is the exact 4-cluster tie Q(0,1) == Q(2,3), decided by the last bit of Eigen's vectorised
    # column sums (TreeNJ.cpp:157), which the host restates (distance.cpp eigen_column_sum)
    ("c1.nw_ml.tree", ["-a", "-m"]),
    ("c1.nw_p.tree", ["-a"])])
def test_nw_guide_tree_c1(oracle_build, case, flags):
    assert run_oracle(oracle_build, flags + ["-T", "-i", "0", os.path.join(GOLD, "c1.fa")]) == gold(case)


def test_nw_guide_trees_pdist(oracle_build, tmp_path):
    """Twelve BioNJ trees (5..30 taxa, odd and even) from alignment p-distances: identical newick, including the side of the
    exact final tie that depends on the association and alignment of Eigen's column sums."""
    for c in json.load(open(os.path.join(GOLD, "nw_trees.json"))):
        (tmp_path / "t.fa").write_text(gen.fasta(gen.gen(c["n"], c["L"], c["seed"], sub=c["sub"], indel=c["indel"])))
        assert run_oracle(oracle_build, ["-a", "-T", "-i", "0", str(tmp_path / "t.fa")]) == c["tree"], c


def test_default_flow_with_tree_reestimation(oracle_build):
    """`--fasta -a` / `-a -T` with the default two rounds of guide-tree re-estimation (main.cpp:404-430,
    DistanceFactoryPrealigned): byte-identical FASTA and newick for 8 x 120, md5 for 64 x 400."""
    c1 = os.path.join(GOLD, "c1.fa")
    assert run_oracle(oracle_build, ["--fasta", "-a", c1]) == gold("c1.a_iter.out.fa")
    assert run_oracle(oracle_build, ["-a", "-T", c1]) == gold("c1.a_iter.tree")
    md5 = json.load(open(os.path.join(GOLD, "md5.json")))
    c2 = os.path.join(GOLD, "c2.fa")
    assert hashlib.md5(run_oracle(oracle_build, ["--fasta", "-a", c2]).encode()).hexdigest() == md5["c2.a_iter.out.fa"]
    assert hashlib.md5(run_oracle(oracle_build, ["-a", "-T", c2]).encode()).hexdigest() == md5["c2.a_iter.tree"]


def test_nw_guide_tree_c2(oracle_build):
    assert run_oracle(oracle_build, ["-a", "-m", "-T", "-i", "0", os.path.join(GOLD, "c2.fa")]) == gold("c2.nw_ml.tree")


@pytest.mark.parametrize("case,flags", [("c1.cs.out.fa", []), ("c1.cs_ml.out.fa", ["-m"])])
def test_csprofile_alignment(oracle_build, case, flags):
    out = run_oracle(oracle_build, ["--fasta"] + flags + ["--tree", os.path.join(GOLD, "c1.tree"), "--cs_profile",
                                                         os.path.join(GOLD, "K50.lib"), os.path.join(GOLD, "c1.fa")])
    assert out == gold(case)


def test_c3_256x1000_md5(oracle_build, tmp_path):
    """BASELINE config 3 family (256 x 1000 aa, --mldist): byte-identical FASTA, pinned by md5."""
    md5 = json.load(open(os.path.join(GOLD, "md5.json")))
    fa = gen.fasta(gen.gen(256, 1000, 3))
    assert hashlib.md5(fa.encode()).hexdigest() == md5["c3.fa"]
    (tmp_path / "c3.fa").write_text(fa)
    out = run_oracle(oracle_build, ["--fasta", "-m", "-t", os.path.join(GOLD, "c3.tree"), str(tmp_path / "c3.fa")])
    assert hashlib.md5(out.encode()).hexdigest() == md5["c3.out.fa"]


def test_c4_128x1000_codons_md5(oracle_build, tmp_path):
    """BASELINE config 4 at full size (128 x 1000 codons, --codon, one pass on the reference's guide tree): ~20 s of oracle."""
    md5 = json.load(open(os.path.join(GOLD, "md5.json")))
    (tmp_path / "c4.fa").write_text(gen.fasta(gen.gen_codon(128, 1000, 4)))
    out = run_oracle(oracle_build, ["--codon", "--fasta", "-t", os.path.join(GOLD, "c4.tree"), str(tmp_path / "c4.fa")])
    assert hashlib.md5(out.encode()).hexdigest() == md5["c4.out.fa"]


@pytest.mark.parametrize("cs", [False, True])
def test_c5_1024x600_md5(oracle_build, tmp_path, cs):
    """BASELINE config 5 at full size (1024 x 600 aa, one pass): ~30 s of oracle; with the synthetic K = 4000 library
    (createProfile of 1024 leaves: ~2 min of oracle) only when PGM_SLOW_TESTS=1."""
    if cs and os.environ.get("PGM_SLOW_TESTS") != "1":
        pytest.skip("2 minutes of CPU: set PGM_SLOW_TESTS=1 (verified in round 2: md5 8c4438f7... identical to the reference binary)")
    md5 = json.load(open(os.path.join(GOLD, "md5.json")))
    (tmp_path / "c5.fa").write_text(gen.fasta(gen.gen(1024, 600, 6)))
    args = ["--fasta", "-t", os.path.join(GOLD, "c5.tree")]
    if cs:
        (tmp_path / "K4000syn.lib").write_text(gen.genlib(4000, 11))
        args += ["--cs_profile", str(tmp_path / "K4000syn.lib")]
    out = run_oracle(oracle_build, args + [str(tmp_path / "c5.fa")])
    assert hashlib.md5(out.encode()).hexdigest() == md5["c5.cs.out.fa" if cs else "c5.out.fa"]


@pytest.mark.parametrize("case", ["cd1", "cd2"])
def test_codon_fasta_identical_to_reference(oracle_build, case):
    """--codon: 61-state ECM model, codon alphabet (BASELINE config 4 in small)."""
    out = run_oracle(oracle_build, ["--codon", "--fasta", "-t", os.path.join(GOLD, case + ".tree"), os.path.join(GOLD, case + ".fa")])
    assert out == gold(case + ".out.fa")


@pytest.mark.parametrize("name,flags", [("c2.cs.out.fa", ["--cs_profile", os.path.join(GOLD, "K50.lib")]), ("c2.m.out.fa", ["-m"])])
def test_c2_variants_md5(oracle_build, name, flags):
    md5 = json.load(open(os.path.join(GOLD, "md5.json")))
    out = run_oracle(oracle_build, ["--fasta"] + flags + ["--tree", os.path.join(GOLD, "c2.tree"), os.path.join(GOLD, "c2.fa")])
    assert hashlib.md5(out.encode()).hexdigest() == md5[name]


@pytest.mark.parametrize("case,n,L,seed,sub,indel", [("cd3", 40, 330, 23, 0.05, 0.008), ("cd4", 64, 500, 24, 0.04, 0.005)])
def test_codon_family_md5(oracle_build, tmp_path, case, n, L, seed, sub, indel):
    """Larger codon families: md5 of the reference's FASTA (pins the oracle's 61-state path at scale)."""
    md5 = json.load(open(os.path.join(GOLD, "md5.json")))
    fa = gen.fasta(gen.gen_codon(n, L, seed, sub=sub, indel=indel))
    assert hashlib.md5(fa.encode()).hexdigest() == md5[case + ".fa"]
    (tmp_path / "cd.fa").write_text(fa)
    out = run_oracle(oracle_build, ["--codon", "--fasta", "-t", os.path.join(GOLD, case + ".tree"), str(tmp_path / "cd.fa")])
    assert hashlib.md5(out.encode()).hexdigest() == md5[case + ".out.fa"]


def test_codon_nw_distance_pairs(oracle_build, tmp_path):
    nw = json.load(open(os.path.join(GOLD, "nw_pairs_codon.json")))
    for seed, p in nw.items():
        (tmp_path / "p.fa").write_text(p["fasta"])
        assert run_oracle(oracle_build, ["--codon", "-a", "-m", "-T", "-i", "0", str(tmp_path / "p.fa")]) == p["ml"], seed


def test_tandem_repeat_families(oracle_build, tmp_path):
    """The repeat-edge branch of the hot function (PredIterator's repeat arm, Graph.h:232-238; markAlternativePath,
    GraphAlign.h:165-198; n_tr_indels) pinned against the reference binary: `--fasta -R --read_repeats` on families with
    annotated tandem repeats — FASTA and the "TR indels" lines per internal node identical; two families also through the
    default flow (guide-tree re-estimation in between)."""
    cases = json.load(open(os.path.join(GOLD, "repeats.json")))
    assert sum(c["changes_alignment"] for c in cases) >= 3 and any("1" in ln.split(": ")[-1] for c in cases for ln in c["tr_lines"][:-1])
    exe = os.path.join(oracle_build, "pgmsa_oracle")
    for c in cases:
        seqs, trd = gen.gen_repeat_family(c["n"], c["L"], c["seed"], annotate_every=c["annotate_every"])
        (tmp_path / "r.fa").write_text(gen.fasta(seqs)); (tmp_path / "r.trd").write_text(trd); (tmp_path / "r.tree").write_text(c["tree"])
        r = subprocess.run([exe, "--fasta", "-R", "--read_repeats", str(tmp_path / "r.trd"), "-t", str(tmp_path / "r.tree"), str(tmp_path / "r.fa")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == c["out"], c["seed"]
        assert [ln for ln in r.stderr.splitlines() if "TR indels" in ln] == c["tr_lines"], c["seed"]
        if "out_default_flow" in c:
            assert run_oracle(oracle_build, ["--fasta", "-a", "--read_repeats", str(tmp_path / "r.trd"), str(tmp_path / "r.fa")]) == c["out_default_flow"], c["seed"]


def _early_args(rec):
    """arguments of an early.json record with the files of tests/golden spelled out"""
    out = []
    for x in rec["args"]:
        out.append(os.path.join(GOLD, x) if x.endswith(".tree") or x.endswith(".lib") else x)
    return out + ["--early_refinement", os.path.join(GOLD, rec["input"])]


def test_early_refinement(oracle_build, tmp_path):
    """--early_refinement (ProgressiveAlignment.h:102-243): the second call site of alignGraphs (:170) — the graph of an internal
    node against the graphs of its grandchildren, not cleaned — plus mergeGraphsIncremental, Graph::reset / rmNodes: the reference
    binary's FASTA on the committed families (fixed tree, --mldist, context-specific profiles, both default flows, codons, 64 x 400
    as md5) and on four repeat families (with their "TR indels" lines)."""
    cases = json.load(open(os.path.join(GOLD, "early.json")))
    assert sum(c["changes_alignment"] for k, c in cases.items() if k != "repeats") >= 5
    for name, c in cases.items():
        if name == "repeats":
            continue
        out = run_oracle(oracle_build, _early_args(c))
        if "md5" in c:
            assert hashlib.md5(out.encode()).hexdigest() == c["md5"], name
        else:
            assert out == c["out"], name
    exe = os.path.join(oracle_build, "pgmsa_oracle")
    for c in cases["repeats"]:
        seqs, trd = gen.gen_repeat_family(c["n"], c["L"], c["seed"], annotate_every=c["annotate_every"])
        (tmp_path / "r.fa").write_text(gen.fasta(seqs)); (tmp_path / "r.trd").write_text(trd); (tmp_path / "r.tree").write_text(c["tree"])
        r = subprocess.run([exe, "--fasta", "-R", "--read_repeats", str(tmp_path / "r.trd"), "--early_refinement", "-t", str(tmp_path / "r.tree"), str(tmp_path / "r.fa")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert r.stdout == c["out"], c["seed"]
        assert [ln for ln in r.stderr.splitlines() if "TR indels" in ln] == c["tr_lines"], c["seed"]


def test_default_flow_without_nwdist(oracle_build, tmp_path):
    """The reference's default flow from sequences alone, WITHOUT -a (initial distances: DistanceFactoryAngle): BASELINE config 1 /
    config 2 inputs as worded (`--fasta c2.fa`), a codon family; the 256 x 1000 family with --mldist behind PGM_SLOW_TESTS."""
    md5 = json.load(open(os.path.join(GOLD, "md5.json")))
    assert run_oracle(oracle_build, ["--fasta", os.path.join(GOLD, "c1.fa")]) == gold("c1.default.out.fa")
    assert run_oracle(oracle_build, ["--codon", "--fasta", os.path.join(GOLD, "cd1.fa")]) == gold("cd1.default.out.fa")
    assert hashlib.md5(run_oracle(oracle_build, ["--fasta", os.path.join(GOLD, "c2.fa")]).encode()).hexdigest() == md5["c2.default.out.fa"]
    if os.environ.get("PGM_SLOW_TESTS"):
        (tmp_path / "c3.fa").write_text(gen.fasta(gen.gen(256, 1000, 3)))
        assert hashlib.md5(run_oracle(oracle_build, ["--fasta", "--mldist", str(tmp_path / "c3.fa")]).encode()).hexdigest() == md5["c3.default_m.out.fa"]


def test_angle_guide_trees_and_the_exact_tie(oracle_build, tmp_path):
    """Twelve families of the default flow WITHOUT -a (k-mer angle distances, DistanceFactoryAngle.h:100-105): the initial guide tree
    (-T -i 0) and the final alignment, byte for byte.  Every BioNJ run ends in an exact tie that the last bits of the distances decide;
    what it took (round 4; 9 of 12 trees before): the cosine matrix is not symmetric in its last bits (rows scaled before the product,
    columns after it) and BioNJ reads it by rows, by columns and at (i, j) as well as (j, i) — the reference's orientation; the depth
    blocks of Eigen's GEMM (L1d / 128 = 384 terms: two blocks for the 400 amino-acid 2-mers); `/ 1.4` a true division; log and exp
    correctly rounded like those of the glibc 2.17 the binary carries.  A campaign of 120 further random families against the reference
    binary itself (tools/diff_campaign.py --angle) found no difference."""
    for c in json.load(open(os.path.join(GOLD, "angle_trees.json"))):
        (tmp_path / "k.fa").write_text(gen.fasta(gen.gen(c["n"], c["L"], c["seed"], sub=c["sub"], indel=c["indel"])))
        assert run_oracle(oracle_build, ["-T", "-i", "0", str(tmp_path / "k.fa")]) == c["tree"], c["seed"]
        assert run_oracle(oracle_build, ["--fasta", str(tmp_path / "k.fa")]) == c["fasta"], c["seed"]


def test_large_guide_trees_from_angle_distances(oracle_build, tmp_path):
    """The committed guide trees of BASELINE configs 4 and 5 are the reference's own (`-T -i 0`: k-mer angle distances, BioNJ, midpoint
    root; tests/golden/make_golden.py full_size): 128 codon sequences, and 1024 sequences — 1021 joins, the criterion scanned on the host
    threads from 512 clusters on (ranges of columns combined in order: Eigen's first minimum), no copy of the matrix per join."""
    (tmp_path / "c4.fa").write_text(gen.fasta(gen.gen_codon(128, 1000, 4)))
    assert run_oracle(oracle_build, ["--codon", "-T", "-i", "0", str(tmp_path / "c4.fa")]) == gold("c4.tree")
    (tmp_path / "c5.fa").write_text(gen.fasta(gen.gen(1024, 600, 6)))
    assert run_oracle(oracle_build, ["-T", "-i", "0", str(tmp_path / "c5.fa")]) == gold("c5.tree")
