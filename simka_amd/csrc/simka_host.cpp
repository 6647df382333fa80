// simka_host.cpp -- host half of libsimka_hip.so: the N x N finalisation that the reference also
// runs on the driver (SimkaDistance, O(N^2) work), the CSV writer and the read packer.
//
//   distance formulas  ref: src/core/SimkaDistance.cpp:920-1226
//   matrix builders    ref: src/core/SimkaDistance.hpp:155-475 (zero diagonal, i<j loops, float32 cells)
//   output order/names ref: src/core/SimkaDistance.cpp:617-647
//   CSV                ref: src/core/SimkaDistance.cpp:653-699
//
// Arithmetic widths follow the reference expression by expression (double / long double /
// one float32 temporary), because the golden CSVs pin the float32 cell printed with "%f".
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <algorithm>
#include <vector>
#include <zlib.h>

#include "../../include/simka_hip.h"

#define SIMKA_EXPORT extern "C" __attribute__((visibility("default")))

namespace {

// what one pair (i<j) contributes to the formulas
struct PairTerms {
    double a, b, c;            // shared distinct, distinct only in i, only in j   (get_abc, .cpp:920-926)
    uint64_t Di, Dj, Ni, Nj;   // per-sample totals
    uint64_t Sij, Sji, bc, shared;
    uint64_t chord, hell, whit, canb;
    long double normi, normj;  // _chord_sqrt_N2
    long double kl;
};

enum Family { PRESENCE, ABUNDANCE, SIMPLE, COMPLEX };
struct MatrixDef {
    const char *name;
    Family family;
    bool asym;
    double (*fn)(const PairTerms &, bool swapped);
};

// swapped == true evaluates the (j,i) cell of an asymmetric matrix
double pa_chord(const PairTerms &t, bool) {                       // .cpp:1117-1127
    const double p = sqrt((t.a + t.b) * (t.a + t.c));
    return p == 0 ? sqrt(2) : sqrt(2 * (1 - t.a / p));
}
double pa_whittaker(const PairTerms &t, bool) {                   // .cpp:1129-1145
    if (t.a + t.b == 0 || t.a + t.c == 0) return 1;
    const double fb = t.b / (t.a + t.b), fc = t.c / (t.a + t.c), gb = t.a / (t.a + t.b), gc = t.a / (t.a + t.c);
    return 0.5 * (fb + fc + fabs(gb - gc));
}
double pa_kulczynski(const PairTerms &t, bool) {                  // .cpp:1156-1170
    if (t.a + t.b == 0 || t.a + t.c == 0) return 1;
    return 1 - 0.5 * (t.a / (t.a + t.b) + t.a / (t.a + t.c));
}
double pa_braycurtis(const PairTerms &t, bool) {                  // .cpp:1172-1183
    const double den = 2 * t.a + t.b + t.c;
    return den == 0 ? 1 : (t.b + t.c) / den;
}
double pa_jaccard(const PairTerms &t, bool) {                     // .cpp:1200-1208
    const double den = t.a + t.b + t.c;
    return den == 0 ? 1 : (t.b + t.c) / den;
}
double pa_simka_jaccard(const PairTerms &t, bool) {               // .cpp:1215-1218
    const double num = 2 * t.shared, den = t.Di + t.Dj;
    return den == 0 ? 1 : 1 - num / den;
}
double pa_simka_jaccard_asym(const PairTerms &t, bool sw) {       // .cpp:1219-1225
    const double num = t.shared, den = sw ? t.Dj : t.Di;
    return den == 0 ? 1 : 1 - num / den;
}
double pa_ochiai(const PairTerms &t, bool) {                      // .cpp:1188-1198 (float32 temporary)
    const float root = sqrt((t.a + t.b) * (t.a + t.c));
    return root == 0 ? 1 : 1 - (t.a / root);
}
double ab_simka_jaccard(const PairTerms &t, bool) {               // .cpp:1041-1065 symmetrical
    const double num = (double)t.Sij + (double)t.Sji, den = (double)t.Ni + (double)t.Nj;
    return den == 0 ? 1 : 1 - num / den;
}
double ab_simka_jaccard_asym(const PairTerms &t, bool sw) {       // asymmetrical
    const double num = sw ? t.Sji : t.Sij, den = sw ? t.Nj : t.Ni;
    return den == 0 ? 1 : 1 - num / den;
}
double ab_ochiai(const PairTerms &t, bool) {                      // .cpp:1068-1078
    const double A1 = t.Sij, B1 = t.Sji, A0 = t.Ni, B0 = t.Nj;
    if (A0 == 0 || B0 == 0) return 1;
    return 1 - sqrt(A1 / A0) * sqrt(B1 / B0);
}
double ab_sorensen(const PairTerms &t, bool) {                    // .cpp:1081-1096
    const double A1 = t.Sij, B1 = t.Sji, A0 = t.Ni, B0 = t.Nj;
    const double den = A0 * B1 + A1 * B0;
    return den == 0 ? 1 : 1 - (2 * A1 * B1) / den;
}
double ab_jaccard(const PairTerms &t, bool) {                     // .cpp:1099-1115
    const double A1 = t.Sij, B1 = t.Sji, A0 = t.Ni, B0 = t.Nj;
    const double den = A0 * B1 + A1 * B0 - A1 * B1;
    return den == 0 ? 1 : 1 - (A1 * B1) / den;
}
double ab_braycurtis(const PairTerms &t, bool) {                  // .cpp:928-939
    const double uni = (double)(t.Ni + t.Nj);
    if (uni == 0) return 1;
    const double inter = (double)(2 * t.bc);
    return 1 - inter / uni;
}
double ab_chord(const PairTerms &t, bool) {                       // .cpp:942-959
    const double den = (double)(t.normi * t.normj);
    if (den == 0) return sqrt(2);
    const long double r = sqrtl(2 - 2 * (long double)t.chord / den);
    return (double)r;
}
double ab_hellinger(const PairTerms &t, bool) {                   // .cpp:962-972
    const double uni = sqrt((double)t.Ni) * sqrt((double)t.Nj);
    if (uni == 0) return sqrt(2);
    const double inter = (double)(2 * t.hell);
    return sqrt(2 - (inter / uni));
}
double ab_kulczynski(const PairTerms &t, bool) {                  // .cpp:1024-1038; kul[j][i] is never written
    if (t.Ni == 0 || t.Nj == 0) return 1;
    const long double n1 = (double)t.bc / (double)t.Ni;           // _kulczynski_minNiNj[i][j] == _brayCurtisNumerator
    const long double n2 = (double)0 / (double)t.Nj;
    return (double)(1 - 0.5 * (n1 + n2));
}
double ab_whittaker(const PairTerms &t, bool) {                   // .cpp:988-998
    const long double uni = (long double)(t.Ni * t.Nj);
    if (uni == 0) return 1;
    return (double)(0.5 * ((long double)t.whit / uni));
}
double ab_jensenshannon(const PairTerms &t, bool) {               // .cpp:1001-1007
    if (t.kl == 0) return 1;
    return (double)sqrtl(0.5 * t.kl);
}
double ab_canberra(const PairTerms &t, bool) {                    // .cpp:1010-1021
    const double den = t.a + t.b + t.c;
    return den == 0 ? 1 : (1 / den) * (double)t.canb;
}

const MatrixDef MATRICES[] = {
    { "mat_presenceAbsence_chord", PRESENCE, false, pa_chord },
    { "mat_presenceAbsence_whittaker", PRESENCE, false, pa_whittaker },
    { "mat_presenceAbsence_kulczynski", PRESENCE, false, pa_kulczynski },
    { "mat_presenceAbsence_braycurtis", PRESENCE, false, pa_braycurtis },
    { "mat_presenceAbsence_jaccard", PRESENCE, false, pa_jaccard },
    { "mat_presenceAbsence_simka-jaccard", PRESENCE, false, pa_simka_jaccard },
    { "mat_presenceAbsence_simka-jaccard_asym", PRESENCE, true, pa_simka_jaccard_asym },
    { "mat_presenceAbsence_ochiai", PRESENCE, false, pa_ochiai },
    { "mat_abundance_simka-jaccard", ABUNDANCE, false, ab_simka_jaccard },
    { "mat_abundance_simka-jaccard_asym", ABUNDANCE, true, ab_simka_jaccard_asym },
    { "mat_abundance_ab-ochiai", ABUNDANCE, false, ab_ochiai },
    { "mat_abundance_ab-sorensen", ABUNDANCE, false, ab_sorensen },
    { "mat_abundance_ab-jaccard", ABUNDANCE, false, ab_jaccard },
    { "mat_abundance_braycurtis", ABUNDANCE, false, ab_braycurtis },
    { "mat_abundance_jaccard", ABUNDANCE, false, nullptr },   // derived from the float Bray-Curtis matrix (.hpp:463-475)
    { "mat_abundance_chord", SIMPLE, false, ab_chord },
    { "mat_abundance_hellinger", SIMPLE, false, ab_hellinger },
    { "mat_abundance_kulczynski", SIMPLE, false, ab_kulczynski },
    { "mat_abundance_whittaker", COMPLEX, false, ab_whittaker },
    { "mat_abundance_jensenshannon", COMPLEX, false, ab_jensenshannon },
    { "mat_abundance_canberra", COMPLEX, false, ab_canberra },
};
const int NB_MATRICES = (int)(sizeof(MATRICES) / sizeof(MATRICES[0]));
const int IDX_BRAYCURTIS = 13, IDX_AB_JACCARD = 14;

void fill_matrix(const simka_stats_view &v, const MatrixDef &def, float *out) {
    const uint64_t n = v.nb_samples;
    uint64_t cell = 0;
    for (uint64_t i = 0; i < n; i++) {
        for (uint64_t j = i + 1; j < n; j++, cell++) {
            PairTerms t;
            memset(&t, 0, sizeof t);
            t.Di = v.nb_distinct[i]; t.Dj = v.nb_distinct[j]; t.Ni = v.nb_kmers[i]; t.Nj = v.nb_kmers[j];
            t.shared = v.distinct_shared[cell];
            t.a = (double)t.shared; t.b = (double)(t.Di - t.shared); t.c = (double)(t.Dj - t.shared);
            t.Sij = v.shared_ij[cell]; t.Sji = v.shared_ji[cell]; t.bc = v.bray_curtis[cell];
            if (v.chord) { t.chord = v.chord[cell]; t.hell = v.hellinger[cell]; }
            t.normi = (long double)sqrt((double)v.sum_sq[i]);     // sqrt(strtoull(..)) -> double, stored long double (.cpp:139)
            t.normj = (long double)sqrt((double)v.sum_sq[j]);
            if (v.whittaker) { t.whit = v.whittaker[cell]; t.canb = v.canberra[cell]; t.kl = (long double)v.kl[cell]; }
            const double dij = def.fn(t, false);
            out[i * n + j] = (float)dij;
            out[j * n + i] = (float)(def.asym ? def.fn(t, true) : dij);
        }
    }
}

}  // namespace

SIMKA_EXPORT int simka_nb_matrices(void) { return NB_MATRICES; }
SIMKA_EXPORT const char *simka_matrix_name(int which) { return (which < 0 || which >= NB_MATRICES) ? nullptr : MATRICES[which].name; }
SIMKA_EXPORT int simka_matrix_enabled(int which, uint32_t flags) {
    if (which < 0 || which >= NB_MATRICES) return 0;
    switch (MATRICES[which].family) {
        case SIMPLE: return (flags & SIMKA_DIST_SIMPLE) ? 1 : 0;     // .cpp:637-642
        case COMPLEX: return (flags & SIMKA_DIST_COMPLEX) ? 1 : 0;   // .cpp:644-648
        default: return 1;
    }
}

SIMKA_EXPORT int simka_compute_matrix(const simka_stats_view *v, int which, float *out) {
    if (!v || !out || which < 0 || which >= NB_MATRICES) return SIMKA_ERR_INVALID;
    if (!simka_matrix_enabled(which, v->dist_flags)) return SIMKA_ERR_INVALID;
    const uint64_t n = v->nb_samples;
    memset(out, 0, n * n * sizeof(float));
    if (which == IDX_AB_JACCARD) {
        std::vector<float> bray(n * n, 0.f);
        fill_matrix(*v, MATRICES[IDX_BRAYCURTIS], bray.data());
        for (uint64_t c = 0; c < n * n; c++) { const double B = bray[c]; out[c] = (float)((2 * B) / (1 + B)); }
        return SIMKA_OK;
    }
    fill_matrix(*v, MATRICES[which], out);
    return SIMKA_OK;
}

SIMKA_EXPORT int simka_write_matrix_csv(const char *dir, const char *name, const char *const *ids, uint32_t n,
                                        const float *m, int gz) {
    if (!dir || !name || !ids || !m) return SIMKA_ERR_INVALID;
    std::string text;
    text.reserve((size_t)n * n * 10 + 64);
    for (uint32_t i = 0; i < n; i++) { text += ';'; text += ids[i]; }
    text += '\n';
    char cell[64];
    for (uint32_t i = 0; i < n; i++) {
        text += ids[i];
        for (uint32_t j = 0; j < n; j++) {
            snprintf(cell, sizeof cell, ";%f", (double)m[(size_t)i * n + j]);   // Stringify::format("%f", float)
            text += cell;
        }
        text += '\n';
    }
    const std::string path = std::string(dir) + "/" + name + ".csv" + (gz ? ".gz" : "");
    if (gz) {
        gzFile g = gzopen(path.c_str(), "wb");
        if (!g) return SIMKA_ERR_IO;
        // (gzwrite takes an unsigned length and returns an int: a matrix of ~15000 samples passes 2 GiB of text)
        size_t done = 0;
        bool ok = true;
        while (ok && done < text.size()) {
            const size_t chunk = std::min<size_t>(text.size() - done, (size_t)1 << 28);
            const int w = gzwrite(g, text.data() + done, (unsigned)chunk);
            if (w <= 0 || (size_t)w != chunk) ok = false;
            done += chunk;
        }
        if (gzclose(g) != Z_OK || !ok) return SIMKA_ERR_IO;
    } else {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) return SIMKA_ERR_IO;
        const size_t w = fwrite(text.data(), 1, text.size(), f);
        if (fclose(f) != 0 || w != text.size()) return SIMKA_ERR_IO;
    }
    return SIMKA_OK;
}

// 2-bit packer.  Letters other than ACGT (either case) end the current fragment.
SIMKA_EXPORT int64_t simka_pack_read(const char *seq, uint64_t len, uint64_t *packed, uint64_t *nb_bases, uint64_t *offsets_out) {
    if (!seq || !packed || !nb_bases || !offsets_out) return -1;
    static const int8_t CODE[256] = {
#define R16 -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1
        R16, R16, R16, R16,
        /* 0x40 */ -1, 0, -1, 1, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1, -1, -1,
        /* 0x50 */ -1, -1, -1, -1, 2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
        /* 0x60 */ -1, 0, -1, 1, -1, -1, -1, 3, -1, -1, -1, -1, -1, -1, -1, -1,
        /* 0x70 */ -1, -1, -1, -1, 2, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
        R16, R16, R16, R16, R16, R16, R16, R16
#undef R16
    };
    uint64_t pos = *nb_bases;
    int64_t nfrag = 0;
    bool open = false;
    for (uint64_t i = 0; i < len; i++) {
        const int c = CODE[(unsigned char)seq[i]];
        if (c < 0) { open = false; continue; }
        if (!open) { offsets_out[nfrag++] = pos; open = true; }
        const uint64_t w = pos >> 5; const unsigned sh = (unsigned)(pos & 31) * 2;
        if (sh == 0) packed[w] = 0;
        packed[w] |= (uint64_t)c << sh;
        pos++;
    }
    *nb_bases = pos;
    return nfrag;
}
