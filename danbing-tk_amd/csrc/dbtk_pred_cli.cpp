// danbing-tk-pred, the command line (host C++ over include/dbtk_pred.h): same arguments, same three output files as the
// reference's src/pred.cpp:14-84.  The cohort's count vectors go to the GPU a few samples at a time; the matrix stays in HBM.
//   danbing-tk-pred <INPUT1: trkmc.ar files + read depths> <INPUT2: ikmer.meta> <OUTPUT1: raw matrix> <OUTPUT2: corrected> <OUTPUT3: bias.tsv>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <string>
#include <vector>

#include "../../include/dbtk_pred.h"

static void die(const std::string& m, int code = 1) { fprintf(stderr, "%s\n", m.c_str()); exit(code); }

// save_matrix (pred.h:236-249): the low 4 bytes of rows and columns, then the column-major float32 data
static void save_matrix(const std::string& fn, const float* d, uint64_t nrow, uint64_t ncol) {
    printf("saving matrix to %s\n", fn.c_str());
    FILE* f = fopen(fn.c_str(), "wb");
    if (!f) die("cannot create " + fn);
    const uint32_t r = (uint32_t)nrow, c = (uint32_t)ncol;
    bool ok = fwrite(&r, 4, 1, f) == 1 && fwrite(&c, 4, 1, f) == 1 && fwrite(d, 4, nrow * ncol, f) == nrow * ncol;
    if (fclose(f) || !ok) die("write error on " + fn);
    printf("matrix dim: (%llu,%llu) size: %llu bytes\n", (unsigned long long)nrow, (unsigned long long)ncol, (unsigned long long)(nrow * ncol * 4));
}

int main(int argc, char** argv) {
    if (argc < 2) {
        fprintf(stderr, "\nUsage: danbing-tk-pred <INPUT1> <INPUT2> <OUTPUT1> <OUTPUT2> <OUTPUT3>\n"
                        "INPUT1      metadata of *.trkmc.ar files, consisting of 2 columns.\n"
                        " col1       *.trkmc.ar file name\n"
                        " col2       read depth\n"
                        "INPUT2      invariant kmers of an RPGG build\n"
                        "OUTPUT1     raw genotype matrix. Row: sample. Column: kmer.\n"
                        "OUTPUT2     bias-corrected genotype matrix. Row: sample. Column: kmer.\n"
                        "OUTPUT3     bias matrix. Row: sample. Column: TR locus.\n"
                        "MI355X build:\n"
                        "  --device <INT>  GPU to use [0]\n\n");
        return 0;
    }
    int argi = 1, device = 0;
    while (argi < argc && argv[argi][0] == '-') {
        const std::string a = argv[argi];
        if (a == "--device" && argi + 1 < argc) { device = atoi(argv[argi + 1]); argi += 2; }
        else if (a == "-f" && argi + 1 < argc) argi += 2;  // developer flag of the reference (its body is commented out there): accepted, ignored
        else die("invalid option: " + a);
    }
    if (argc - argi < 5) die("expected 5 file arguments");
    const std::string finGtMeta = argv[argi], finIkMeta = argv[argi + 1], foutRaw = argv[argi + 2], fout = argv[argi + 3], foutBias = argv[argi + 4];
    printf("metadata of *.trkmc.ar: %s\ninvariant kmers: %s\nraw genotype matrix will be written to: %s\n"
           "bias-corrected genotype matrix will be written to: %s\nbias matrix will be written to: %s\n",
           finGtMeta.c_str(), finIkMeta.c_str(), foutRaw.c_str(), fout.c_str(), foutBias.c_str());
    // read_gt_meta (pred.h:41-49): name <tab> read depth per line
    std::vector<std::string> fns;
    std::vector<float> rds;
    {
        std::ifstream fin(finGtMeta);
        if (!fin) die("cannot open " + finGtMeta);
        std::string f1, f2;
        while (std::getline(fin, f1, '\t') && std::getline(fin, f2)) { fns.push_back(f1); rds.push_back(std::stof(f2)); }
    }
    const uint64_t ns = fns.size();
    if (!ns) die(finGtMeta + ": no samples");
    dbtk_pred_t* P = nullptr;
    if (dbtk_pred_create_from_file(device, ns, finIkMeta.c_str(), &P)) die(dbtk_last_error());
    const uint64_t nk = dbtk_pred_nk(P), ntr = dbtk_pred_ntr(P);
    printf("%llu loci in total.\n", (unsigned long long)ntr);
    printf("reading %llu gt files\n", (unsigned long long)ns);
    const uint64_t B = 16;  // samples per transfer
    std::vector<uint64_t> buf(B * nk);
    for (uint64_t s0 = 0; s0 < ns; s0 += B) {
        const uint64_t n = std::min<uint64_t>(B, ns - s0);
        for (uint64_t i = 0; i < n; ++i) {  // load_eachBinGT (pred.h:166-186): 8 bytes (nk) | 8 * nk bytes (counts)
            FILE* f = fopen(fns[s0 + i].c_str(), "rb");
            if (!f) die("cannot open " + fns[s0 + i], 134);
            uint64_t nkf = 0;
            if (fread(&nkf, 8, 1, f) != 1 || nkf != nk) { fprintf(stderr, "nk %llu != nk_ %llu\n", (unsigned long long)nkf, (unsigned long long)nk); exit(134); }  // the reference asserts
            if (fread(buf.data() + i * nk, 8, nk, f) != nk) die("truncated " + fns[s0 + i], 134);
            fclose(f);
        }
        if (dbtk_pred_load_samples(P, s0, n, buf.data(), rds.data() + s0)) die(dbtk_last_error());
    }
    std::vector<float> mat(ns * nk);
    printf("normalizaing read depth\n");
    if (dbtk_pred_matrix(P, mat.data())) die(dbtk_last_error());
    save_matrix(foutRaw, mat.data(), ns, nk);
    printf("computing/correcting bias\n");
    if (dbtk_pred_correct(P)) die(dbtk_last_error());
    float ms[3];
    dbtk_pred_times(P, ms);
    printf("finished in %.3f ms on the GPU (bias sums %.3f, bias normalisation %.3f, correction %.3f)\n", ms[0] + ms[1] + ms[2], ms[0], ms[1], ms[2]);
    if (dbtk_pred_matrix(P, mat.data())) die(dbtk_last_error());
    save_matrix(fout, mat.data(), ns, nk);
    std::vector<float> bias(ns * ntr);
    if (dbtk_pred_bias(P, bias.data())) die(dbtk_last_error());
    {   // save_matrix with the tsv format (pred.cpp:51, pred.h:251-258): rows = samples, tab-separated, default stream precision, no final newline
        printf("saving matrix to %s\n", foutBias.c_str());
        FILE* f = fopen(foutBias.c_str(), "w");
        if (!f) die("cannot create " + foutBias);
        std::string line;
        char num[64];
        for (uint64_t s = 0; s < ns; ++s) {
            line.clear();
            for (uint64_t t = 0; t < ntr; ++t) {
                snprintf(num, sizeof num, "%g", (double)bias[t * ns + s]);
                if (t) line += '\t';
                line += num;
            }
            if (s + 1 < ns) line += '\n';
            fwrite(line.data(), 1, line.size(), f);
        }
        if (fclose(f)) die("write error on " + foutBias);
    }
    dbtk_pred_free(P);
    return 0;
}
