#!/usr/bin/env python
"""Generates the committed fixtures under tests/golden/.  Run in the BUILD container only
(it imports the reference's Python simulator from /root/reference; nothing under /root/reference
exists on the GPU box and nothing here is read from it at test time).

  sim_gaussian_30k.cov.gz / .bin   configs[0]: docs/hmm_test recipe (docs/hmm_test/README.md:99-136) with the
                                   reference's own programs/src/simulate_coverage_data.py, truth parameter tables
                                   re-typed from docs/hmm_test/README.md:43-57,68-80 (they are absent from the snapshot)
  sim_truth_emission.tsv / sim_truth_transition.tsv   those tables
  sim100k_{gaussian,exp_gaussian,negative_binomial}.cov.gz   the doc's full recipe (100 000 observations, contigs of 80 000 and
                                   20 000 bases, regionChangeRate 0.001) for the three emission families it names; the
                                   gaussian truth table is the doc's, the other two (programs/tests/test_files/simulate_coverage/
                                   is absent from the snapshot) are written here in the simulator's format with the same
                                   means / variances; sim100k_truth_*.tsv hold them
  cfg1_*.                          configs[1] outputs of the ORACLE (self-regression, not a reference pin)
  small_em_*                       a 3-contig, 2-region EM run of the ORACLE (self-regression)
  chunks_creator_test_1*.cov       the reference loader test's own data files (programs/tests/test_files/chunks_creator)
"""
import gzip
import os
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference/programs"

EMISSION_TSV = """#State\tDistribution\tComponents\tParameter\tValues_Region_0\tValues_Region_1
Err\tGaussian\t1\tMean\t2.0\t3.0
Err\tGaussian\t1\tVar\t2.4\t4.0
Err\tGaussian\t1\tWeight\t1.0\t1.0
Dup\tGaussian\t1\tMean\t10.0\t15.0
Dup\tGaussian\t1\tVar\t12.0\t20.0
Dup\tGaussian\t1\tWeight\t1.0\t1.0
Hap\tGaussian\t1\tMean\t20.0\t30.0
Hap\tGaussian\t1\tVar\t24.0\t40.0
Hap\tGaussian\t1\tWeight\t1.0\t1.0
Col\tGaussian\t4\tMean\t40.0,60.0,80.0,100.0\t60.0,90.0,120.0,150.0
Col\tGaussian\t4\tVar\t48.0,72.0,96.0,120.0\t80.0,120.0,160.0,200.0
Col\tGaussian\t4\tWeight\t0.4,0.3,0.2,0.1\t0.5,0.4,0.05,0.05
"""
TRANSITION_TSV = """#Region\tState\tErr\tDup\tHap\tCol\tEnd
0\tErr\t0.8999\t0.01\t0.08\t0.01\t1.0e-4
0\tDup\t0.01\t0.8999\t0.08\t0.01\t1.0e-4
0\tHap\t0.001\t0.005\t0.9899\t0.004\t1.0e-4
0\tCol\t0.02\t0.02\t0.06\t0.8999\t1.0e-4
0\tStart\t2.50e-1\t2.50e-01\t2.50e-01\t2.50e-01\t0.0
1\tErr\t0.8999\t0.04\t0.05\t0.01\t1.0e-4
1\tDup\t0.01\t0.8999\t0.08\t0.01\t1.0e-4
1\tHap\t0.003\t0.003\t0.9899\t0.004\t1.0e-4
1\tCol\t0.02\t0.03\t0.05\t0.8999\t1.0e-4
1\tStart\t2.50e-1\t2.50e-01\t2.50e-01\t2.50e-01\t0.0
"""


def simulate():
    sys.path.insert(0, os.path.join(REF, "src"))
    import simulate_coverage_data as sim   # the reference's simulator
    open(os.path.join(HERE, "sim_truth_emission.tsv"), "w").write(EMISSION_TSV)
    open(os.path.join(HERE, "sim_truth_transition.tsv"), "w").write(TRANSITION_TSV)
    np.random.seed(1234)                   # seed = 1234 + cfg 0
    emis = sim.parseEmissionParametersPerRegion(os.path.join(HERE, "sim_truth_emission.tsv"))
    trans = sim.parseTransitionMatrixPerRegion(os.path.join(HERE, "sim_truth_transition.tsv"))
    n, contigs = 30000, [24000, 6000]
    regions, states, obs = sim.generateObservations(trans, emis, numberOfObservations=n, regionChangeRate=0.001,
                                                    alphaMatrix=None)
    region_cov = [p[2]["Mean"][0] for p in emis]
    cov = os.path.join(HERE, "sim_gaussian_30k.cov")
    sim.writeObservationsIntoCov(regions, states, obs, region_cov, contigs, pathToWrite=cov)
    with open(cov, "rb") as f, gzip.GzipFile(cov + ".gz", "wb", mtime=0) as g:
        shutil.copyfileobj(f, g)
    os.unlink(cov)
    # the same observations as windowLen 1 / chunkLen 1000 `.bin` chunks (one base = one window, so no averaging)
    from flagger_amd import synth
    store = synth.WindowStore(
        cov=np.zeros(0, np.uint16), mapq=np.zeros(0, np.uint16), clip=np.zeros(0, np.uint16), annot=np.zeros(0, np.uint64),
        truth=np.zeros(0, np.int8), prediction=np.zeros(0, np.int8), chunk_off=np.zeros(1, np.int64), chunk_ctg=[],
        chunk_ctg_len=np.zeros(0, np.int32), chunk_s=np.zeros(0, np.int32), chunk_e=np.zeros(0, np.int32),
        window_len=1, chunk_len=1000, region_coverages=[int(x) for x in region_cov], avg_alignment_len=0,
        annotation_names=("no_annotation", "whole_genome", "TEST_CONTIG_0", "TEST_CONTIG_1"))
    covs, annots, truths, ctgs, cl, cs, ce, off = [], [], [], [], [], [], [], [0]
    start = 0
    for ci, L in enumerate(contigs):
        for s, e in synth.chunk_bounds(L, 1000):
            sl = slice(start + s, start + e + 1)
            c = np.minimum(250, np.asarray(obs[sl], dtype=np.float64)).astype(np.uint16)
            r = np.asarray(regions[sl], dtype=np.uint64)
            covs.append(c)
            annots.append((np.uint64(1) << np.uint64(0)) | (np.uint64(1) << np.uint64(1 + ci)) | (r << np.uint64(58)))  # bit = index-1, ptBlock.c:225-228
            truths.append(np.asarray(states[sl], dtype=np.int8))
            ctgs.append(f"TEST_CONTIG_{ci}"); cl.append(L); cs.append(s); ce.append(e); off.append(off[-1] + (e - s + 1))
        start += L
    store.cov = np.concatenate(covs); store.mapq = np.zeros_like(store.cov); store.clip = np.zeros_like(store.cov)
    store.annot = np.concatenate(annots); store.truth = np.concatenate(truths)
    store.prediction = np.full(store.cov.size, -1, np.int8); store.chunk_off = np.asarray(off, np.int64)
    store.chunk_ctg = ctgs; store.chunk_ctg_len = np.asarray(cl, np.int32)
    store.chunk_s = np.asarray(cs, np.int32); store.chunk_e = np.asarray(ce, np.int32)
    store.write_bin(os.path.join(HERE, "sim_gaussian_30k.bin"))
    print("simulated", store.n_windows, "windows in", store.n_chunks, "chunks")


FAMILIES = {
    "gaussian": EMISSION_TSV,
    "exp_gaussian": EMISSION_TSV.replace("Err\tGaussian\t1\tMean\t2.0\t3.0\nErr\tGaussian\t1\tVar\t2.4\t4.0\nErr\tGaussian\t1\tWeight\t1.0\t1.0\n",
                                         "Err\tTruncated Exponential\t1\tMean\t2.0\t3.0\nErr\tTruncated Exponential\t1\tTrunc_Point\t5.0\t7.5\n"),
    "negative_binomial": EMISSION_TSV.replace("Gaussian", "Negative Binomial"),
}


def simulate_families():
    """docs/hmm_test/README.md:99-136 at full size for every emission family the simulator knows."""
    sys.path.insert(0, os.path.join(REF, "src"))
    import simulate_coverage_data as sim
    for k, (name, table) in enumerate(FAMILIES.items()):
        ep = os.path.join(HERE, f"sim100k_truth_emission_{name}.tsv")
        open(ep, "w").write(table)
        np.random.seed(4321 + k)
        emis = sim.parseEmissionParametersPerRegion(ep)
        trans = sim.parseTransitionMatrixPerRegion(os.path.join(HERE, "sim_truth_transition.tsv"))
        regions, states, obs = sim.generateObservations(trans, emis, numberOfObservations=100000, regionChangeRate=0.001, alphaMatrix=None)
        cov = os.path.join(HERE, f"sim100k_{name}.cov")
        sim.writeObservationsIntoCov(regions, states, obs, [p[2]["Mean"][0] for p in emis], [80000, 20000], pathToWrite=cov)
        with open(cov, "rb") as f, gzip.GzipFile(cov + ".gz", "wb", mtime=0) as g:
            shutil.copyfileobj(f, g)
        os.unlink(cov)
        print(name, os.path.getsize(cov + ".gz"), "bytes")


def oracle_runs():
    from flagger_amd import synth
    orc = os.path.join(ROOT, "oracle", "hf_oracle")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    np.savetxt(os.path.join(HERE, "alpha_hifi.tsv"), synth.HIFI_ALPHA, fmt="%.3f", delimiter="\t")
    cases = {
        "cfg1": (synth.config(1), ["-n", "0", "-W", "4000", "-A", os.path.join(HERE, "alpha_hifi.tsv")]),
        "small_em": (synth.synthesize([1_500_000, 600_000, 90_000], 2000, 500_000, [20, 30], seed=21,
                                      region_run_bases=(20_000, 300_000)),
                     ["-n", "8", "-W", "2000", "-A", os.path.join(HERE, "alpha_hifi.tsv"), "-x", "ont-r10"]),
    }
    for name, (store, args) in cases.items():
        binp = os.path.join(HERE, f"{name}.bin")
        store.write_bin(binp)
        out = os.path.join(HERE, f"{name}_expected")
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out)
        subprocess.run([orc, "-i", binp, "-o", out, "-@", "4"] + args, check=True)
        print(name, store.n_windows, "windows ->", sorted(os.listdir(out)))


def loader_files():
    src = os.path.join(REF, "tests", "test_files", "chunks_creator")
    for f in ("test_1.cov", "test_1_with_labels.cov", "test_1.cov.gz"):
        shutil.copy(os.path.join(src, f), os.path.join(HERE, "chunks_creator_" + f))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "families":
        simulate_families()
        sys.exit(0)
    simulate()
    simulate_families()
    oracle_runs()
    loader_files()
