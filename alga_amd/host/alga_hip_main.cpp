// alga_amd/host/alga_hip_main.cpp -- `alga_hip`: ALGA's command line in front of the MI355X overlap engine.
//
//   alga_hip --file1=reads.fasta [--file2=mates.fasta] --output=contigs.fasta [--threads=N] [--error_rate=R | --error-rate=R]
//            [--serialize=1] [-l MINOVERLAP] [--rsoemo=N] [--scale=F] [--retl=N --retr=N] [--remove_reads_with_n=0|1] [--rna=0|1]
//            [--device=K] [--gpus=N | --gpu-list=0,1,2,...] [--alga=/path/to/stock/ALGA]
//
// --gpus=N: the overlap graph on the GPUs K .. K+N-1 of this node (alga_multi_*, include/alga_amd.h: one host thread and one engine
// per GPU, keys and edge lists exchanged over RCCL / xGMI) -- the counterpart of the reference's --threads for this stage
// (src/Params.cpp:237-294).  Every GPU runs the input stage on the files itself (each over its own PCIe link), so no node set
// crosses xGMI.  --gpu-list names the devices explicitly; a device named more than once selects the copy transport (testing).
//
// It reads the input exactly like the reference (src/IO/InputReader.cpp, src/IO/ReadPreprocess.cpp, src/main.cpp:93-266),
// builds the overlap graph on the GPU and writes `<TEST_NAME>_beforeSimplifier.graph` in the reference's own dump
// format (src/DataStructures/Graph.cpp:269-297).  Stock ALGA started with the same arguments plus
// --deserialize_graph=1 loads that file instead of running its GraphCreator (src/main.cpp:242) and carries on with
// the unchanged simplifier / contig stages; `--alga=` does that hand-off in one go.
// Both spellings of the error-rate option are accepted (the reference registers `error_rate` only, src/Params.cpp:226).
#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "GraphCreatorHIP.hpp"
#include "ingest.hpp"

static bool opt(const char *arg, const char *name, std::string &val) {
    size_t n = strlen(name);
    if (strncmp(arg, name, n) == 0 && arg[n] == '=') { val = arg + n + 1; return true; }
    return false;
}

int main(int argc, char **argv) {
    using clk = std::chrono::steady_clock;
    std::string file1, file2, output, alga_exe, v;
    alga_host::IngestParams ip;
    double error_rate = 0.0;
    int device = 0, serialize = 1, gpus = 1;
    std::vector<int32_t> gpu_list;
    std::vector<std::string> passthrough;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (opt(a, "--file1", v)) file1 = v;
        else if (opt(a, "--file2", v)) file2 = v;
        else if (opt(a, "--output", v)) output = v;
        else if (opt(a, "--threads", v)) ip.threads = std::max(1, atoi(v.c_str()));
        else if (opt(a, "--error_rate", v) || opt(a, "--error-rate", v) || opt(a, "--er", v)) error_rate = atof(v.c_str());
        else if (opt(a, "--serialize", v)) serialize = atoi(v.c_str());
        else if (opt(a, "--rsoemo", v)) ip.rsoemo = atoi(v.c_str());
        else if (opt(a, "--scale", v)) ip.scale = (float) atof(v.c_str());
        else if (opt(a, "--retl", v) || opt(a, "--read_end_trim_left", v)) ip.trim_left = atoi(v.c_str());
        else if (opt(a, "--retr", v) || opt(a, "--read_end_trim_right", v)) ip.trim_right = atoi(v.c_str());
        else if (opt(a, "--remove_reads_with_n", v)) ip.remove_reads_with_n = atoi(v.c_str());
        else if (opt(a, "--rna", v)) ip.rna = atoi(v.c_str());
        else if (opt(a, "--device", v)) device = atoi(v.c_str());
        else if (opt(a, "--gpus", v)) gpus = std::max(1, atoi(v.c_str()));
        else if (opt(a, "--gpu-list", v)) { gpu_list.clear(); for (size_t k = 0; k < v.size();) { gpu_list.push_back(atoi(v.c_str() + k)); size_t c = v.find(',', k); if (c == std::string::npos) break; k = c + 1; } }
        else if (opt(a, "--alga", v)) alga_exe = v;
        else if (!strcmp(a, "-l") && i + 1 < argc) ip.min_overlap = atoi(argv[++i]);
        else { fprintf(stderr, "alga_hip: unrecognized option '%s'\n", a); return 2; }
        // the hand-off to stock ALGA drops the error-rate option: the supplement it switches on (src/Params.cpp:357-359) has
        // already run here and nothing downstream reads the rate
        // ... and --serialize / --deserialize_graph: the hand-off always goes through the dump this program writes
        const bool is_er = !strncmp(a, "--error_rate", 12) || !strncmp(a, "--error-rate", 12) || !strncmp(a, "--er=", 5);
        const bool is_ser = !strncmp(a, "--serialize", 11) || !strncmp(a, "--deserialize_graph", 19);
        if (strncmp(a, "--device", 8) && strncmp(a, "--alga", 6) && strncmp(a, "--gpus", 6) && strncmp(a, "--gpu-list", 10) && !is_er && !is_ser) { passthrough.push_back(a); if (!strcmp(a, "-l")) passthrough.push_back(argv[i]); }
    }
    if (file1.empty()) { fprintf(stderr, "\nERROR - PLEASE PROVIDE THE INPUT FILE using --file1 option!\n"); return 1; }
    if (output.empty()) { fprintf(stderr, "\nERROR - PLEASE PROVIDE THE OUTPUT FILE NAME!\n"); return 1; }
    const std::string graph = alga_host::test_name(file1, ip.scale, ip.remove_reads_with_n) + "_beforeSimplifier.graph";
    if (!alga_exe.empty()) {
        // The hand-off is the dump: it is always written, and a file of that name left behind by an earlier run must not be
        // what stock ALGA loads if this run fails half-way (src/main.cpp:241-242 falls back to its own CPU creator without one).
        serialize = 1;
        (void) unlink(graph.c_str());
    }
    auto t0 = clk::now();
    // Input stages (src/IO/InputReader.cpp, src/IO/ReadPreprocess.cpp, src/main.cpp:93-266) on the GPU: the host maps the files and
    // moves their bytes, the node set stays in HBM for the graph creator.  Inputs that stage does not take (file types other than
    // FASTA / FASTQ, random replacement of N) are parsed on the host cores and join the GPU at the duplicate / prefix removal.
    if (gpu_list.empty()) for (int k = 0; k < gpus; k++) gpu_list.push_back(device + k);
    const int n_ranks = (int) gpu_list.size();
    alga_multi *multi = nullptr;
    alga_engine *engine = nullptr;
    if (n_ranks > 1) {
        bool distinct = true;
        for (int a = 0; a < n_ranks; a++) for (int b = 0; b < a; b++) distinct = distinct && gpu_list[(size_t) a] != gpu_list[(size_t) b];
        int rc = alga_multi_create(gpu_list.data(), n_ranks, distinct ? ALGA_TRANSPORT_AUTO : ALGA_TRANSPORT_COPY, &multi);
        if (rc == ALGA_ERR_UNSUPPORTED) rc = alga_multi_create(gpu_list.data(), n_ranks, ALGA_TRANSPORT_COPY, &multi);     // no RCCL on this box
        if (rc != ALGA_OK) { fprintf(stderr, "alga_amd: cannot open %d HIP devices (status %d)\n", n_ranks, rc); return 1; }
        engine = alga_multi_engine(multi, 0);
    } else if (alga_engine_create(gpu_list[0], &engine) != ALGA_OK) { fprintf(stderr, "alga_amd: no usable HIP device\n"); return 1; }
    const auto t_engine = clk::now();
    alga_device_node_set nodes;
    alga_host::Parsed parsed;
    alga_ingest_params cp;
    alga_ingest_default_params(&cp);
    cp.trim_left = ip.trim_left; cp.trim_right = ip.trim_right; cp.remove_reads_with_n = ip.remove_reads_with_n; cp.rna = ip.rna; cp.scale = ip.scale;
    cp.min_overlap = ip.min_overlap; cp.rsoemo = ip.rsoemo; cp.remove_pref_reads = ip.remove_pref_reads; cp.threads = ip.threads;
    alga_ingest_info info;
    auto t1 = t_engine;
    // the input stage on every GPU of the run, side by side (rank 0's on this thread): the node set never crosses xGMI
    std::vector<alga_device_node_set> rank_nodes((size_t) n_ranks);
    std::vector<alga_ingest_info> rank_info((size_t) n_ranks);
    std::vector<int> rank_rc((size_t) n_ranks, ALGA_OK);
    {
        std::vector<std::thread> th;
        auto ingest = [&](int r) {
            alga_engine *er = multi ? alga_multi_engine(multi, r) : engine;
            rank_rc[(size_t) r] = alga_ingest_device(er, file1.c_str(), file2.empty() ? nullptr : file2.c_str(), &cp, &rank_nodes[(size_t) r], &rank_info[(size_t) r]);
        };
        for (int r = 1; r < n_ranks; r++) th.emplace_back(ingest, r);
        ingest(0);
        for (std::thread &x : th) x.join();
    }
    int irc = ALGA_OK;
    for (int r = 0; r < n_ranks; r++) if (rank_rc[(size_t) r] != ALGA_OK) { irc = rank_rc[(size_t) r]; if (irc != ALGA_ERR_UNSUPPORTED) { fprintf(stderr, "%s\n", alga_last_error(multi ? alga_multi_engine(multi, r) : engine)); return 1; } }
    nodes = rank_nodes[0]; info = rank_info[0];
    if (irc == ALGA_OK) {
        fprintf(stderr, "device ingest: upload %.1f ms, lines + records %.1f ms, duplicate/prefix removal %.1f ms (wall)\n", info.ms_upload,
                info.ms_parse - info.ms_upload, info.ms_preprocess);
        parsed.records = info.records; parsed.removed_n = info.removed_n; parsed.removed_str = info.removed_str;
        parsed.min_overlap = info.min_overlap; parsed.rsoemo = info.rsoemo; parsed.li_kmer_length = info.li_kmer_length;
        t1 = t_engine + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double, std::milli>(info.ms_parse));
    } else {                                               // ALGA_ERR_UNSUPPORTED: the host parser, then the duplicate / prefix removal on every GPU
        std::string err = alga_host::parse(file1, file2, ip, parsed);
        if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
        t1 = clk::now();
        alga_preprocess_input pin{parsed.rows.data(), parsed.W, parsed.len.data(), (int64_t) (2 * parsed.R), ip.remove_pref_reads, 3 + parsed.li_kmer_length};
        for (int r = 0; r < n_ranks; r++) {
            alga_engine *er = multi ? alga_multi_engine(multi, r) : engine;
            if (alga_preprocess_nodes(er, &pin, &rank_nodes[(size_t) r]) != ALGA_OK) { fprintf(stderr, "%s\n", alga_last_error(er)); return 1; }
        }
        nodes = rank_nodes[0];
    }
    auto t1b = clk::now();
    fprintf(stderr, "input read: %lld records -> %d nodes (removed: %d with N, %d STR, %d duplicate/prefix, %d too short)\n",
            (long long) parsed.records, nodes.n, parsed.removed_n, parsed.removed_str, nodes.removed_prefix, nodes.removed_short);
    fprintf(stderr, "MIN_OVERLAP_PREF_SUF: %d\nREMOVE_SMALL_OVERLAP_EDGES_MIN_OVERLAP: %d\n", parsed.min_overlap, parsed.rsoemo);
    fprintf(stderr, "Creating GraphCreator\n");
    alga_prefsuf_stats st;
    const alga_edge *d_final = nullptr;
    uint64_t n_final = 0;
    alga_host::GraphCreatorPrefSufHIP creator(engine, nodes.d_words, nodes.stride_words, nodes.d_len, nodes.n, parsed.min_overlap, parsed.rsoemo);
    if (multi) {
        std::vector<alga_nodes> per_rank;
        for (int r = 0; r < n_ranks; r++) per_rank.push_back(alga_nodes{rank_nodes[(size_t) r].d_words, rank_nodes[(size_t) r].stride_words, rank_nodes[(size_t) r].d_len, rank_nodes[(size_t) r].n, nullptr, nullptr});
        alga_prefsuf_params pp;
        alga_prefsuf_default_params(&pp);
        pp.min_overlap = parsed.min_overlap; pp.rsoe_min_overlap = parsed.rsoemo;
        int rc = alga_multi_prefsuf_build_device(multi, per_rank.data(), &pp, &d_final, &n_final);
        if (rc != ALGA_OK) { fprintf(stderr, "alga_amd: %s (status %d)\n", alga_multi_last_error(multi), rc); return 1; }
        alga_multi_stats ms;
        std::vector<alga_prefsuf_stats> rs((size_t) n_ranks);
        alga_multi_last_stats(multi, &ms, rs.data());
        st = rs[0];
        fprintf(stderr, "%d GPUs (%s): keys %.1f ms, key all-gather %.1f ms, build %.1f ms, edge gather %.1f ms (rank 0's host clock)%s\n", n_ranks,
                ms.transport == ALGA_TRANSPORT_RCCL ? "RCCL" : "peer copies", ms.ms_keys, ms.ms_share, ms.ms_build, ms.ms_gather,
                ms.fell_back_to_one_gpu ? "; the source-side form does not take this input: rank 0 built the graph alone" : "");
    } else {
        creator.startAlignmentGraphCreation();
        st = creator.stats();
        d_final = creator.deviceEdges();
        n_final = creator.countEdges();
    }
    auto t2 = clk::now();
    if (error_rate > 0.01) {                                                   // src/Params.cpp:358-359, src/main.cpp:300-355
        fprintf(stderr, "Before supplement, G has %llu edges\n", (unsigned long long) n_final);
        std::vector<int32_t> hl((size_t) nodes.n);
        if (nodes.n && alga_copy_to_host(engine, hl.data(), nodes.d_len, hl.size() * sizeof(int32_t)) != ALGA_OK) { fprintf(stderr, "alga_amd: cannot read node lengths back\n"); return 1; }
        double sum = 0; long long cnt = 0;
        for (int32_t l : hl) if (l > 0) { sum += l; cnt++; }
        alga_pkb_params pp;
        alga_pkb_derive_params(cnt ? sum / (double) cnt : 0.0, ip.scale, error_rate, parsed.li_kmer_length, &pp);
        alga_nodes nd{nodes.d_words, nodes.stride_words, nodes.d_len, nodes.n, nullptr, nullptr};
        int rc;
        if (multi) {
            // the k-mer groups dealt out over the GPUs by hash; every rank ends with the same graph, rank 0's comes back
            std::vector<alga_nodes> per_rank;
            for (int r = 0; r < n_ranks; r++) per_rank.push_back(alga_nodes{rank_nodes[(size_t) r].d_words, rank_nodes[(size_t) r].stride_words, rank_nodes[(size_t) r].d_len, rank_nodes[(size_t) r].n, nullptr, nullptr});
            rc = alga_multi_pkb_supplement_device(multi, per_rank.data(), &pp, d_final, n_final, &d_final, &n_final);
            if (rc != ALGA_OK) { fprintf(stderr, "alga_amd: %s (status %d)\n", alga_multi_last_error(multi), rc); return 1; }
        } else {
            rc = alga_pkb_supplement_device(engine, &nd, &pp, d_final, n_final, nullptr, &d_final, &n_final);
            if (rc != ALGA_OK) { fprintf(stderr, "alga_amd: %s (status %d)\n", alga_last_error(engine), rc); return 1; }
        }
        fprintf(stderr, "After supplement G has %llu edges\n", (unsigned long long) n_final);
    }
    std::vector<alga_edge> final_edges((size_t) n_final);
    if (n_final && alga_copy_to_host(engine, final_edges.data(), d_final, final_edges.size() * sizeof(alga_edge)) != ALGA_OK) { fprintf(stderr, "alga_amd: cannot read the edges back\n"); return 1; }
    fprintf(stderr, "Before first simplifier graph has %llu edges\n", (unsigned long long) n_final);
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    fprintf(stderr, "HIP start-up %.1f ms, parse %.1f ms, duplicate/prefix removal %.1f ms wall (device %.3f ms), overlap graph %.1f ms wall (device %.3f ms: seed %.3f probe %.3f group %.3f reduce %.3f emit %.3f)\n",
            ms(t0, t_engine), ms(t_engine, t1), ms(t1, t1b), nodes.ms_device, ms(t1b, t2), st.ms_total, st.ms_seed, st.ms_probe, st.ms_group, st.ms_reduce, st.ms_emit);
    if (serialize) {
        int rc = alga_write_graph(graph.c_str(), nodes.n, final_edges.data(), n_final);
        if (rc != ALGA_OK) { fprintf(stderr, "cannot write %s\n", graph.c_str()); return 1; }
        fprintf(stderr, "Graph serialized! -> %s\n", graph.c_str());
    }
    final_edges.clear(); final_edges.shrink_to_fit();
    creator.clear();
    if (multi) alga_multi_destroy(multi); else alga_engine_destroy(engine);     // the engines and their HBM buffers do not outlive the graph
    if (!alga_exe.empty()) {
        // argv vector, no shell: nothing in a file name is interpreted
        std::vector<std::string> args{alga_exe};
        for (const std::string &a : passthrough) args.push_back(a);
        args.push_back("--deserialize_graph=1");
        std::vector<char *> av;
        std::string shown;
        for (std::string &a : args) { av.push_back(&a[0]); shown += (shown.empty() ? "" : " ") + a; }
        av.push_back(nullptr);
        fprintf(stderr, "handing over to the unchanged simplifier / contig stages: %s\n", shown.c_str());
        pid_t pid = 0;
        extern char **environ;
        if (posix_spawn(&pid, alga_exe.c_str(), nullptr, nullptr, av.data(), environ) != 0) { fprintf(stderr, "alga_hip: cannot start %s\n", alga_exe.c_str()); return 1; }
        int status = 0;
        if (waitpid(pid, &status, 0) < 0) return 1;
        return (WIFEXITED(status) && WEXITSTATUS(status) == 0) ? 0 : 1;
    }
    return 0;
}
