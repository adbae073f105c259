/*
 * cpecan_realign -- command line of the batch front end (include/cpecan_realign.h): the options of cPecanRealign
 * (cPecanRealign.c:17-47, :372-470), cigars on stdin, cigars on stdout.  The cigars are realigned in batches of
 * --batch alignments (one GPU launch each) instead of one at a time.
 */
#define _POSIX_C_SOURCE 200809L
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cpecan_realign.h"

static void usage(void) {
    fprintf(stderr,
            "cpecan_realign [options] seq1[fasta] seq2[fasta] ... < cigars > realigned cigars\n"
            "-l --gapGamma F  -L --matchGamma F  -o --splitMatrixBiggerThanThis N  -r --diagonalExpansion N\n"
            "-t --constraintDiagonalTrim N  -w --alignAmbiguityCharacters (accepted; the reference's aligner never reads it)\n"
            "-x --rescoreOriginalAlignment  -i --rescoreByIdentity  -j --rescoreByPosteriorProb\n"
            "-k --rescoreByIdentityIgnoringGaps  -m --rescoreByPosteriorProbIgnoringGaps  -s --splitIndelsLongerThanThis N\n"
            "-u --outputPosteriorProbs FILE  -z --outputAllPosteriorProbs FILE  -v --outputExpectations FILE\n"
            "-y --loadHmm FILE  -a --logLevel L (ignored)  -b --batch N (most alignments per GPU batch, default 32768;\n"
            "a batch also closes at 64 Mbp of aligned sequence)\n"
            "-g --device N  -G --devices LIST (e.g. 0-7 or 0,2,3: the cigars of every batch are cut into one shard per\n"
            "listed device, run side by side from this process and joined in input order; a device may be listed twice)\n"
            "-h --help\n");
}

/* "0-3", "0,2,5", "1,1": a device list; returns the count or -1 */
static int parse_devices(const char *text, int *out, int cap) {
    int n = 0;
    for (const char *p = text; *p;) {
        char *end;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) return -1;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            if (end == p + 1 || b < a) return -1;
            p = end;
        }
        for (long d = a; d <= b; d++) {
            if (n >= cap) return -1;
            out[n++] = (int)d;
        }
        if (*p == ',') p++;
        else if (*p) return -1;
    }
    return n;
}

static int fail(const char *what) {
    const char *e = cpecan_last_error();
    fprintf(stderr, "cpecan_realign: %s: %s\n", what, e ? e : "");
    return 1;
}

int main(int argc, char **argv) {
    cpecan_realign_options o;
    cpecan_realign_options_default(&o);
    const char *posteriorFile = NULL, *allPosteriorFile = NULL, *expectationsFile = NULL, *hmmFile = NULL;
    long long batch = 32768, device = 0, v;
    int devices[64], nDevices = 0;
    const long long batchBases = 64ll << 20; /* a batch also closes here: ~1.5 GB of anchors on the host */
    static struct option longOpts[] = {{"logLevel", required_argument, 0, 'a'},
                                       {"help", no_argument, 0, 'h'},
                                       {"gapGamma", required_argument, 0, 'l'},
                                       {"matchGamma", required_argument, 0, 'L'},
                                       {"splitMatrixBiggerThanThis", required_argument, 0, 'o'},
                                       {"diagonalExpansion", required_argument, 0, 'r'},
                                       {"constraintDiagonalTrim", required_argument, 0, 't'},
                                       {"alignAmbiguityCharacters", no_argument, 0, 'w'},
                                       {"rescoreOriginalAlignment", no_argument, 0, 'x'},
                                       {"rescoreByIdentity", no_argument, 0, 'i'},
                                       {"rescoreByPosteriorProb", no_argument, 0, 'j'},
                                       {"rescoreByPosteriorProbIgnoringGaps", no_argument, 0, 'm'},
                                       {"rescoreByIdentityIgnoringGaps", no_argument, 0, 'k'},
                                       {"splitIndelsLongerThanThis", required_argument, 0, 's'},
                                       {"outputPosteriorProbs", required_argument, 0, 'u'},
                                       {"outputAllPosteriorProbs", required_argument, 0, 'z'},
                                       {"outputExpectations", required_argument, 0, 'v'},
                                       {"loadHmm", required_argument, 0, 'y'},
                                       {"batch", required_argument, 0, 'b'},
                                       {"device", required_argument, 0, 'g'},
                                       {"devices", required_argument, 0, 'G'},
                                       {0, 0, 0, 0}};
    for (int key; (key = getopt_long(argc, argv, "a:hl:o:r:t:s:wxijkmu:v:y:z:L:b:g:G:", longOpts, NULL)) != -1;) {
        switch (key) {
        case 'a': break;
        case 'h': usage(); return 0;
        case 'l': if (sscanf(optarg, "%f", &o.gapGamma) != 1) return 1; break;
        case 'L': if (sscanf(optarg, "%f", &o.matchGamma) != 1) return 1; break;
        case 'o': if (sscanf(optarg, "%lld", &v) != 1 || v < 0) return 1; o.params.splitMatrixBiggerThanThis = v * v; break;
        case 'r': if (sscanf(optarg, "%lld", &v) != 1) return 1; o.params.diagonalExpansion = v; break;
        case 't': if (sscanf(optarg, "%lld", &v) != 1) return 1; o.constraintDiagonalTrim = v; break;
        case 'w': break;
        case 'x': o.rescoreOriginalAlignment = 1; break;
        case 'i': o.rescoreByIdentity = 1; break;
        case 'j': o.rescoreByPosteriorProb = 1; break;
        case 'k': o.rescoreByIdentityIgnoringGaps = 1; break;
        case 'm': o.rescoreByPosteriorProbIgnoringGaps = 1; break;
        case 's': if (sscanf(optarg, "%lld", &v) != 1 || v < 0) return 1; o.splitIndelsLongerThanThis = v; break;
        case 'u': posteriorFile = optarg; break;
        case 'z': allPosteriorFile = optarg; break;
        case 'v': expectationsFile = optarg; break;
        case 'y': hmmFile = optarg; break;
        case 'b': if (sscanf(optarg, "%lld", &batch) != 1 || batch < 1) return 1; break;
        case 'g': if (sscanf(optarg, "%lld", &device) != 1) return 1; break;
        case 'G': if ((nDevices = parse_devices(optarg, devices, 64)) < 1) { usage(); return 1; } device = devices[0]; break;
        default: usage(); return 1;
        }
    }
    cpecan_model model;
    cpecan_hmm expectations;
    if (hmmFile) { /* cPecanRealign.c:481-486 */
        cpecan_hmm hmm;
        if (cpecan_hmm_load(&hmm, hmmFile) != CPECAN_OK || cpecan_model_from_hmm(&model, &hmm) != CPECAN_OK) return fail("loadHmm");
    } else if (cpecan_model_default(&model, CPECAN_FIVE_STATE) != CPECAN_OK) { /* :489 */
        return fail("model");
    }
    if (expectationsFile && cpecan_hmm_init(&expectations, model.type, 0.000000000001) != CPECAN_OK) return fail("hmm"); /* :497 */
    cpecan_realigner *r = NULL;
    if (cpecan_realigner_create(&r, &model, &o, (int)device) != CPECAN_OK) return fail("options");
    if (nDevices > 1 && cpecan_realigner_set_devices(r, devices, nDevices) != CPECAN_OK) return fail("devices");
    if (optind >= argc) {
        usage();
        return 1;
    }
    for (; optind < argc; optind++)
        if (cpecan_realigner_read_fasta(r, argv[optind]) < 0) return fail(argv[optind]);
    if (cpecan_realigner_set_posterior_files(r, posteriorFile, allPosteriorFile) != CPECAN_OK) return fail("files");

    cpecan_cigar *in = calloc((size_t)batch, sizeof(cpecan_cigar));
    char *line = NULL, *text = NULL;
    size_t lineCap = 0;
    int64_t textCap = 0;
    int status = in ? 0 : 1;
    for (int done = 0; !done && status == 0;) {
        int64_t n = 0;
        long long bases = 0;
        while (n < batch && bases < batchBases) {
            if (getline(&line, &lineCap, stdin) < 0) {
                done = 1;
                break;
            }
            if (line[strspn(line, " \t\r\n")] == 0) continue;
            if (cpecan_cigar_parse(line, &in[n]) != CPECAN_OK) {
                status = fail("cigar");
                break;
            }
            bases += llabs((long long)(in[n].end1 - in[n].start1));
            n++;
        }
        if (status == 0 && n > 0) {
            if (expectationsFile) {
                if (cpecan_realigner_expectations(r, in, n, &expectations) != CPECAN_OK) status = fail("expectations");
            } else {
                cpecan_cigar *out = NULL;
                int64_t nOut = 0;
                if (cpecan_realigner_realign(r, in, n, &out, &nOut) != CPECAN_OK) status = fail("realign");
                for (int64_t i = 0; status == 0 && i < nOut; i++) {
                    const int64_t need = cpecan_cigar_format(&out[i], NULL, 0) + 1;
                    if (need > textCap) {
                        textCap = 2 * need;
                        char *grown = realloc(text, (size_t)textCap);
                        if (!grown) {
                            status = 1;
                            break;
                        }
                        text = grown;
                    }
                    cpecan_cigar_format(&out[i], text, textCap);
                    puts(text);
                }
                cpecan_cigars_free(out, nOut);
            }
        }
        for (int64_t i = 0; i < n; i++) cpecan_cigar_clear(&in[i]);
    }
    if (status == 0 && expectationsFile && cpecan_hmm_write(&expectations, expectationsFile) != CPECAN_OK) status = fail(expectationsFile);
    free(in);
    free(line);
    free(text);
    cpecan_realigner_destroy(r);
    return status;
}
