// rafft_params.h - host side of the energy parameters: ViennaRNA parameter files -> device tables.
//
// The reference gets its energies from ViennaRNA: Glob_parms builds `RNA.md()`, sets `md.temperature = temp`
// and `RNA.fold_compound(sequence, md)` (rafft/utils.py:17-21), i.e. the parameter set currently loaded in
// ViennaRNA (Turner 2004 unless the user called RNA.params_load / read_parameter_file) rescaled to `temp`.
// ViennaRNA is third-party and not in the reference tree; here it is "touched only once up front for the energy
// tables": this file reads the parameter file format ViennaRNA 2.x writes ("## RNAfold parameter file v2.0",
// RNA.params_save / misc/rna_turner2004.par) and restates its temperature rescaling
// (ViennaRNA src/ViennaRNA/params/basic.c, get_scaled_params: G(T) = dH - (dH - G37) * (T + K0) / Tmeasure,
// truncated to int; dangles and multi/exterior mismatches clipped to <= 0 for dangle model 2; loop
// extrapolation lxc scales linearly with T).
//
// Without a loaded file the built-in 37 C tables are used (params/turner2004_tables.h: the published Turner-2004
// model arbitrated by the reference's 11 505 energy rows; no enthalpies, so only temp == 37).
#pragma once
#include <algorithm>
#include "rafft_device.h"
#include "../../params/turner2004_tables.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace rafft_par {

constexpr int NBP = 7;            // ViennaRNA NBPAIRS: CG GC GU UG AU UA NS(non-standard)
constexpr int INF_ = 10000000;    // ViennaRNA INF
constexpr double K0 = 273.15, TMEASURE = 37.0 + K0;

struct SpecialLoop { std::string seq; int e37, dH; };

// One parameter set in ViennaRNA's own array shapes (index 0 of a pair axis = no pair, of a base axis = N).
struct ParamSet {
    bool has_dH = false;
    bool builtin_set = false;          // the compiled-in tables: their rule / model entries are marked in the device tables (scaled_tables)
    std::string source = "built-in Turner 2004, 37 C (params/turner2004_tables.h)";
    int stack[2][NBP + 1][NBP + 1] = {};
    int hairpin[2][31] = {}, bulge[2][31] = {}, interior[2][31] = {};
    int mmH[2][NBP + 1][5][5] = {}, mmI[2][NBP + 1][5][5] = {}, mm1n[2][NBP + 1][5][5] = {}, mm23[2][NBP + 1][5][5] = {},
        mmM[2][NBP + 1][5][5] = {}, mmE[2][NBP + 1][5][5] = {};
    int d5[2][NBP + 1][5] = {}, d3[2][NBP + 1][5] = {};
    int int11[2][NBP + 1][NBP + 1][5][5] = {};
    int int21[2][NBP + 1][NBP + 1][5][5][5] = {};
    int int22[2][NBP + 1][NBP + 1][5][5][5][5] = {};
    int ninio[2] = {0, 0}, max_ninio = 300;
    int ml_base[2] = {0, 0}, ml_closing[2] = {0, 0}, ml_intern[2] = {0, 0};
    int term_au[2] = {0, 0}, duplex_init[2] = {0, 0};
    double lxc = 107.856;
    std::vector<SpecialLoop> tri, tetra, hexa;
};

// the compiled-in set (index [0] = 37 C values; no enthalpies)
inline void builtin(ParamSet &P)
{
    P = ParamSet();
    for (int a = 0; a < 7; a++) for (int b = 0; b < 7; b++) P.stack[0][a][b] = t04_stack[a][b];
    for (int i = 0; i < 31; i++) { P.hairpin[0][i] = t04_hairpin[i]; P.bulge[0][i] = t04_bulge[i]; P.interior[0][i] = t04_interior[i]; }
    for (int t = 0; t < 7; t++) for (int a = 0; a < 5; a++) {
        P.d5[0][t][a] = t04_dangle5[t][a]; P.d3[0][t][a] = t04_dangle3[t][a];
        for (int b = 0; b < 5; b++) {
            P.mmH[0][t][a][b] = t04_mismatch_hairpin[t][a][b]; P.mmI[0][t][a][b] = t04_mismatch_interior[t][a][b];
            P.mm1n[0][t][a][b] = t04_mismatch_interior_1n[t][a][b]; P.mm23[0][t][a][b] = t04_mismatch_interior_23[t][a][b];
            P.mmM[0][t][a][b] = t04_mismatch_multi[t][a][b]; P.mmE[0][t][a][b] = t04_mismatch_exterior[t][a][b];
        }
    }
    for (int t = 0; t < 7; t++) for (int u = 0; u < 7; u++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
        P.int11[0][t][u][a][b] = t04_int11[t][u][a][b];
        for (int c = 0; c < 5; c++) {
            P.int21[0][t][u][a][b][c] = t04_int21[t][u][a][b][c];
            for (int d = 0; d < 5; d++) P.int22[0][t][u][a][b][c][d] = t04_int22[t][u][a][b][c][d];
        }
    }
    P.ninio[0] = T04_NINIO; P.max_ninio = T04_MAX_NINIO;
    P.ml_base[0] = T04_ML_BASE; P.ml_closing[0] = T04_ML_CLOSING; P.ml_intern[0] = T04_ML_INTERN;
    P.term_au[0] = T04_TERMINAL_AU; P.lxc = T04_LXC;
    for (int i = 0; i < T04_N_TRILOOPS; i++) P.tri.push_back({t04_triloops_seq[i], t04_triloops_e[i], 0});
    for (int i = 0; i < T04_N_TETRALOOPS; i++) P.tetra.push_back({t04_tetraloops_seq[i], t04_tetraloops_e[i], 0});
    for (int i = 0; i < T04_N_HEXALOOPS; i++) P.hexa.push_back({t04_hexaloops_seq[i], t04_hexaloops_e[i], 0});
    P.builtin_set = true;
}

// entries of the built-in interior-loop tables that no reference-held energy row exercises (rule / model values)
inline void builtin_unpinned_counts(int out[3])
{
    out[0] = out[1] = out[2] = 0;
    for (int t = 1; t < 7; t++) for (int u = 1; u < 7; u++) for (int a = 1; a < 5; a++) for (int b = 1; b < 5; b++) {
        out[0] += !t04_int11_pinned[t][u][a][b];
        for (int c = 1; c < 5; c++) {
            out[1] += !t04_int21_pinned[t][u][a][b][c];
            for (int e = 1; e < 5; e++) out[2] += !t04_int22_pinned[t][u][a][b][c][e];
        }
    }
}

// ---------------------------------------------------------------- reader

struct Section { std::vector<std::string> tok; std::vector<int> line; int head_line = 0; };   // tokens with the line each came from; line of the '# name' header

inline bool tokenize(const std::string &text, std::map<std::string, Section> &secs, std::string &err)
{
    // strip /* ... */ comments (they may span lines)
    std::string s;
    s.reserve(text.size());
    for (size_t i = 0; i < text.size();) {
        if (text[i] == '/' && i + 1 < text.size() && text[i + 1] == '*') {
            const size_t e = text.find("*/", i + 2);
            if (e == std::string::npos) {
                err = "line " + std::to_string(1 + std::count(text.begin(), text.begin() + i, '\n')) + ": unterminated comment";
                return false;
            }
            for (size_t k = i; k < e + 2; k++) if (text[k] == '\n') s.push_back('\n');     // (line numbers survive)
            i = e + 2;
            s.push_back(' ');
        } else s.push_back(text[i++]);
    }
    bool header = false, ended = false;
    Section *cur = nullptr;
    size_t p = 0;
    int lineno = 0;
    while (p < s.size() && !ended) {
        size_t e = s.find('\n', p);
        if (e == std::string::npos) e = s.size();
        std::string line = s.substr(p, e - p);
        p = e + 1;
        lineno++;
        size_t a = line.find_first_not_of(" \t\r");
        if (a == std::string::npos) continue;
        if (line[a] == '#') {
            if (line.compare(a, 2, "##") == 0) { if (line.find("RNAfold parameter file v2.0") != std::string::npos) header = true; continue; }
            size_t b = line.find_first_not_of(" \t", a + 1);
            if (b == std::string::npos) continue;
            size_t c = line.find_first_of(" \t\r", b);
            std::string name = line.substr(b, c == std::string::npos ? std::string::npos : c - b);
            if (name == "END") { ended = true; break; }
            cur = &secs[name];
            cur->head_line = lineno;
            continue;
        }
        if (!cur) continue;
        size_t q = a;
        while (q < line.size()) {
            size_t b = line.find_first_not_of(" \t\r", q);
            if (b == std::string::npos) break;
            size_t c = line.find_first_of(" \t\r", b);
            cur->tok.push_back(line.substr(b, c == std::string::npos ? std::string::npos : c - b));
            cur->line.push_back(lineno);
            if (c == std::string::npos) break;
            q = c;
        }
    }
    if (!header) { err = "not a ViennaRNA parameter file (no '## RNAfold parameter file v2.0' header)"; return false; }
    return true;
}

inline bool tok_int(const std::string &t, int &v, bool *is_def = nullptr)
{
    if (is_def) *is_def = false;
    if (t == "INF") { v = INF_; return true; }
    if (t == "DEF") { if (is_def) *is_def = true; return is_def != nullptr; }    // "keep the default" marker of parameter files
    char *end = nullptr;
    const double d = strtod(t.c_str(), &end);
    if (end == t.c_str() || *end) return false;
    if (!(d >= -(double)INF_ && d <= (double)INF_)) return false;       // (NaN, infinities, values no int holds: a conversion would be undefined)
    v = (int)d;
    return true;
}

// fill `count` ints from a section; DEF keeps what is there
inline bool fill(const std::map<std::string, Section> &secs, const char *name, std::vector<int *> dst, std::string &err, bool required = true)
{
    auto it = secs.find(name);
    if (it == secs.end()) {
        if (required) { err = std::string("section '# ") + name + "' is missing"; return false; }
        return true;
    }
    const auto &tk = it->second.tok;
    if (tk.size() != dst.size()) {
        // where it goes wrong: the line of the first surplus value, or the last line of a block that is short
        const int at = tk.size() > dst.size() ? it->second.line[dst.size()] : (tk.empty() ? it->second.head_line : it->second.line.back());
        err = std::string("section '# ") + name + "' (line " + std::to_string(it->second.head_line) + "): " + std::to_string(tk.size()) + " values, expected " +
              std::to_string(dst.size()) + (tk.size() > dst.size() ? "; first surplus value on line " : "; block ends on line ") + std::to_string(at);
        return false;
    }
    for (size_t i = 0; i < tk.size(); i++) {
        int v = 0;
        bool keep = false;
        if (!tok_int(tk[i], v, &keep)) { err = std::string("section '# ") + name + "', line " + std::to_string(it->second.line[i]) + ": bad token '" + tk[i] + "'"; return false; }
        if (!keep) *dst[i] = v;
    }
    return true;
}

inline bool parse(const std::string &text, ParamSet &P, std::string &err)
{
    std::map<std::string, Section> secs;
    if (!tokenize(text, secs, err)) return false;
    builtin(P);               // sections a file leaves out, and DEF entries, keep the built-in 37 C values
    // a file must bring enthalpies for everything it brings energies for; start from dH = G37 (temperature independent)
#define DUP(arr) memcpy(P.arr[1], P.arr[0], sizeof P.arr[0])
    DUP(stack); DUP(hairpin); DUP(bulge); DUP(interior); DUP(mmH); DUP(mmI); DUP(mm1n); DUP(mm23); DUP(mmM); DUP(mmE);
    DUP(d5); DUP(d3); DUP(int11); DUP(int21); DUP(int22);
#undef DUP
    P.ninio[1] = P.ninio[0]; P.ml_base[1] = P.ml_base[0]; P.ml_closing[1] = P.ml_closing[0]; P.ml_intern[1] = P.ml_intern[0];
    P.term_au[1] = P.term_au[0];
    bool all_dH = true;
    for (int w = 0; w < 2; w++) {
        const std::string sfx = w ? "_enthalpies" : "";
        const bool req = (w == 0);
        auto has = [&](const char *n) { return secs.count(std::string(n) + sfx) > 0; };
        std::string nm_;
        auto name = [&](const char *n) { nm_ = std::string(n) + sfx; return nm_.c_str(); };
        std::vector<int *> d;
        // # stack: pairs 1..7 x 1..7 (rd_2dim(stack37, NBPAIRS+1, NBPAIRS+1, 1, 1))
        d.clear(); for (int a = 1; a <= NBP; a++) for (int b = 1; b <= NBP; b++) d.push_back(&P.stack[w][a][b]);
        if (w && !has("stack")) all_dH = false;
        if (!fill(secs, name("stack"), d, err, req)) return false;
        struct M3 { const char *n; int (*arr)[NBP + 1][5][5]; };
        const M3 m3[] = {{"mismatch_hairpin", P.mmH}, {"mismatch_interior", P.mmI}, {"mismatch_interior_1n", P.mm1n},
                         {"mismatch_interior_23", P.mm23}, {"mismatch_multi", P.mmM}, {"mismatch_exterior", P.mmE}};
        for (const M3 &m : m3) {       // pairs 1..7, bases 0..4 (rd_3dim(.., NBPAIRS+1, 5, 5, 1, 0, 0))
            d.clear(); for (int t = 1; t <= NBP; t++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) d.push_back(&m.arr[w][t][a][b]);
            if (w && !has(m.n)) all_dH = false;
            if (!fill(secs, name(m.n), d, err, req)) return false;
        }
        d.clear(); for (int t = 1; t <= NBP; t++) for (int a = 0; a < 5; a++) d.push_back(&P.d5[w][t][a]);
        if (w && !has("dangle5")) all_dH = false;
        if (!fill(secs, name("dangle5"), d, err, req)) return false;
        d.clear(); for (int t = 1; t <= NBP; t++) for (int a = 0; a < 5; a++) d.push_back(&P.d3[w][t][a]);
        if (w && !has("dangle3")) all_dH = false;
        if (!fill(secs, name("dangle3"), d, err, req)) return false;
        d.clear(); for (int t = 1; t <= NBP; t++) for (int u = 1; u <= NBP; u++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) d.push_back(&P.int11[w][t][u][a][b]);
        if (w && !has("int11")) all_dH = false;
        if (!fill(secs, name("int11"), d, err, req)) return false;
        d.clear(); for (int t = 1; t <= NBP; t++) for (int u = 1; u <= NBP; u++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) for (int c = 0; c < 5; c++) d.push_back(&P.int21[w][t][u][a][b][c]);
        if (w && !has("int21")) all_dH = false;
        if (!fill(secs, name("int21"), d, err, req)) return false;
        // # int22: pairs 1..6, bases 1..4 only (rd_6dim_slice(.., 1,1,1,1,1,1, 1,1,0,0,0,0)); N and NS entries are derived below
        d.clear(); for (int t = 1; t < NBP; t++) for (int u = 1; u < NBP; u++) for (int a = 1; a < 5; a++) for (int b = 1; b < 5; b++) for (int c = 1; c < 5; c++) for (int e = 1; e < 5; e++) d.push_back(&P.int22[w][t][u][a][b][c][e]);
        if (w && !has("int22")) all_dH = false;
        if (!fill(secs, name("int22"), d, err, req)) return false;
        d.clear(); for (int i = 0; i < 31; i++) d.push_back(&P.hairpin[w][i]);
        if (w && !has("hairpin")) all_dH = false;
        if (!fill(secs, name("hairpin"), d, err, req)) return false;
        d.clear(); for (int i = 0; i < 31; i++) d.push_back(&P.bulge[w][i]);
        if (w && !has("bulge")) all_dH = false;
        if (!fill(secs, name("bulge"), d, err, req)) return false;
        d.clear(); for (int i = 0; i < 31; i++) d.push_back(&P.interior[w][i]);
        if (w && !has("interior")) all_dH = false;
        if (!fill(secs, name("interior"), d, err, req)) return false;
        // update_nst(): entries of the 2x2 table with an N base take the maximum over the concrete bases at that place
        for (int t = 1; t < NBP; t++) for (int u = 1; u < NBP; u++)
            for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) for (int c = 0; c < 5; c++) for (int e = 0; e < 5; e++) {
                if (a && b && c && e) continue;
                int mx = -INF_;
                for (int a2 = (a ? a : 1); a2 <= (a ? a : 4); a2++) for (int b2 = (b ? b : 1); b2 <= (b ? b : 4); b2++)
                    for (int c2 = (c ? c : 1); c2 <= (c ? c : 4); c2++) for (int e2 = (e ? e : 1); e2 <= (e ? e : 4); e2++)
                        mx = std::max(mx, P.int22[w][t][u][a2][b2][c2][e2]);
                P.int22[w][t][u][a][b][c][e] = mx;
            }
    }
    {   // # NINIO: m, m_dH, max
        auto it = secs.find("NINIO");
        if (it != secs.end()) {
            const auto &tk = it->second.tok;
            if (tk.size() < 3) { err = "section '# NINIO': expected 3 values"; return false; }
            if (!tok_int(tk[0], P.ninio[0]) || !tok_int(tk[1], P.ninio[1]) || !tok_int(tk[2], P.max_ninio)) { err = "section '# NINIO': bad token"; return false; }
        } else all_dH = false;
    }
    {   // # ML_params: cu cu_dH cc cc_dH ci ci_dH
        auto it = secs.find("ML_params");
        if (it != secs.end()) {
            const auto &tk = it->second.tok;
            if (tk.size() < 6) { err = "section '# ML_params': expected 6 values"; return false; }
            int *dst[6] = {&P.ml_base[0], &P.ml_base[1], &P.ml_closing[0], &P.ml_closing[1], &P.ml_intern[0], &P.ml_intern[1]};
            for (int i = 0; i < 6; i++) if (!tok_int(tk[i], *dst[i])) { err = "section '# ML_params': bad token"; return false; }
        } else all_dH = false;
    }
    {   // # Misc: DuplexInit dH TerminalAU dH LXC (0)
        auto it = secs.find("Misc");
        if (it != secs.end()) {
            const auto &tk = it->second.tok;
            if (tk.size() < 5) { err = "section '# Misc': expected at least 5 values"; return false; }
            if (!tok_int(tk[0], P.duplex_init[0]) || !tok_int(tk[1], P.duplex_init[1]) || !tok_int(tk[2], P.term_au[0]) || !tok_int(tk[3], P.term_au[1])) { err = "section '# Misc': bad token"; return false; }
            char *end = nullptr;
            P.lxc = strtod(tk[4].c_str(), &end);
            if (end == tk[4].c_str() || !(P.lxc > -1e6 && P.lxc < 1e6)) { err = "section '# Misc': bad LXC"; return false; }
        } else all_dH = false;
    }
    struct SL { const char *n; std::vector<SpecialLoop> *v; size_t len; };
    const SL sl[] = {{"Triloops", &P.tri, 5}, {"Tetraloops", &P.tetra, 6}, {"Hexaloops", &P.hexa, 8}};
    for (const SL &s : sl) {
        auto it = secs.find(s.n);
        if (it == secs.end()) continue;      // (keeps the built-in list, like an absent section in ViennaRNA keeps its defaults)
        const auto &tk = it->second.tok;
        if (tk.size() % 3) { err = std::string("section '# ") + s.n + "': expected (sequence energy enthalpy) triples"; return false; }
        s.v->clear();
        for (size_t i = 0; i < tk.size(); i += 3) {
            SpecialLoop l;
            l.seq = tk[i];
            if (l.seq.size() != s.len || !tok_int(tk[i + 1], l.e37) || !tok_int(tk[i + 2], l.dH)) { err = std::string("section '# ") + s.n + "': bad entry '" + tk[i] + "'"; return false; }
            s.v->push_back(l);
        }
    }
    if (P.tri.size() + P.tetra.size() + P.hexa.size() > 96) { err = "more than 96 special hairpin loops"; return false; }
    P.has_dH = all_dH;
    P.builtin_set = false;             // every entry is the file's (sections it leaves out keep built-in values: they are not marked either)
    return true;
}

// ---------------------------------------------------------------- writer (the layout RNA.params_save produces)

inline std::string format(const ParamSet &P)
{
    std::string o;
    char b[128];
    auto num = [&](int v) { if (v >= INF_) snprintf(b, sizeof b, "   INF"); else snprintf(b, sizeof b, "%6d", v); o += b; };
    static const char *pn[8] = {"NP", "CG", "GC", "GU", "UG", "AU", "UA", "NN"};
    static const char *bn = "NACGU";
    o += "## RNAfold parameter file v2.0\n\n/* written by libraffthip (rafft_save_params); source: " + P.source + " */\n";
    // every energy array is followed by its `_enthalpies` twin, in ViennaRNA's order of sections
    const int nw = P.has_dH ? 2 : 1;
    auto sfx = [](int w) { return std::string(w ? "_enthalpies" : ""); };
    for (int w = 0; w < nw; w++) {
        o += "\n# stack" + sfx(w) + "\n/*  CG     GC     GU     UG     AU     UA     NN  */\n";
        for (int a = 1; a <= NBP; a++) { for (int c = 1; c <= NBP; c++) { num(P.stack[w][a][c]); o += " "; } o += std::string("   /* ") + pn[a] + " */\n"; }
    }
    struct M3 { const char *n; const int (*arr)[NBP + 1][5][5]; };
    const M3 m3[] = {{"mismatch_hairpin", P.mmH}, {"mismatch_interior", P.mmI}, {"mismatch_interior_1n", P.mm1n},
                     {"mismatch_interior_23", P.mm23}, {"mismatch_multi", P.mmM}, {"mismatch_exterior", P.mmE}};
    for (const M3 &m : m3)
        for (int w = 0; w < nw; w++) {
            o += std::string("\n# ") + m.n + sfx(w) + "\n";
            for (int t = 1; t <= NBP; t++) for (int a = 0; a < 5; a++) {
                for (int c = 0; c < 5; c++) { num(m.arr[w][t][a][c]); o += " "; }
                snprintf(b, sizeof b, "   /* %s,%c */\n", pn[t], bn[a]); o += b;
            }
        }
    for (int k = 0; k < 2; k++)
        for (int w = 0; w < nw; w++) {
            o += std::string("\n# ") + (k ? "dangle3" : "dangle5") + sfx(w) + "\n/*   N      A      C      G      U  */\n";
            for (int t = 1; t <= NBP; t++) { for (int a = 0; a < 5; a++) { num((k ? P.d3 : P.d5)[w][t][a]); o += " "; } o += std::string("   /* ") + pn[t] + " */\n"; }
        }
    for (int w = 0; w < nw; w++) {
        o += "\n# int11" + sfx(w) + "\n";
        for (int t = 1; t <= NBP; t++) for (int u = 1; u <= NBP; u++) {
            snprintf(b, sizeof b, "/* %s..%s */\n", pn[t], pn[u]); o += b;
            for (int a = 0; a < 5; a++) { for (int c = 0; c < 5; c++) { num(P.int11[w][t][u][a][c]); o += " "; } o += "\n"; }
        }
    }
    for (int w = 0; w < nw; w++) {
        o += "\n# int21" + sfx(w) + "\n";
        for (int t = 1; t <= NBP; t++) for (int u = 1; u <= NBP; u++) for (int a = 0; a < 5; a++) {
            snprintf(b, sizeof b, "/* %s.%c..%s */\n", pn[t], bn[a], pn[u]); o += b;
            for (int c = 0; c < 5; c++) { for (int e = 0; e < 5; e++) { num(P.int21[w][t][u][a][c][e]); o += " "; } o += "\n"; }
        }
    }
    for (int w = 0; w < nw; w++) {
        o += "\n# int22" + sfx(w) + "\n";
        for (int t = 1; t < NBP; t++) for (int u = 1; u < NBP; u++) for (int a = 1; a < 5; a++) for (int c = 1; c < 5; c++) {
            snprintf(b, sizeof b, "/* %s.%c%c..%s */\n", pn[t], bn[a], bn[c], pn[u]); o += b;
            for (int e = 1; e < 5; e++) { for (int f = 1; f < 5; f++) { num(P.int22[w][t][u][a][c][e][f]); o += " "; } o += "\n"; }
        }
    }
    const int (*lin[3])[31] = {P.hairpin, P.bulge, P.interior};
    const char *ln[3] = {"hairpin", "bulge", "interior"};
    for (int k = 0; k < 3; k++)
        for (int w = 0; w < nw; w++) {
            o += std::string("\n# ") + ln[k] + sfx(w) + "\n";
            for (int i = 0; i < 31; i++) { num(lin[k][w][i]); o += ((i % 10) == 9 || i == 30) ? "\n" : " "; }
        }
    o += "\n# NINIO\n/* Ninio = MIN(max, m*|n1-n2| */\n/*       m   m_dH     max  */\n";
    snprintf(b, sizeof b, "%6d %6d %6d\n", P.ninio[0], P.ninio[1], P.max_ninio); o += b;
    o += "\n# ML_params\n/* F = cu*n_unpaired + cc + ci*loop_degree (+TermAU) */\n/*      cu  cu_dH     cc  cc_dH     ci  ci_dH  */\n";
    snprintf(b, sizeof b, "%6d %6d %6d %6d %6d %6d\n", P.ml_base[0], P.ml_base[1], P.ml_closing[0], P.ml_closing[1], P.ml_intern[0], P.ml_intern[1]); o += b;
    o += "\n# Misc\n/* all parameters are pairs of 'energy enthalpy' */\n/*    DuplexInit     TerminalAU      LXC */\n";
    snprintf(b, sizeof b, "%6d %6d %6d %6d %12.6f %6d\n", P.duplex_init[0], P.duplex_init[1], P.term_au[0], P.term_au[1], P.lxc, 0); o += b;
    struct SL { const char *n; const std::vector<SpecialLoop> *v; };
    const SL sl[] = {{"Hexaloops", &P.hexa}, {"Tetraloops", &P.tetra}, {"Triloops", &P.tri}};
    for (const SL &s : sl) {
        o += std::string("\n# ") + s.n + "\n";
        for (const SpecialLoop &l : *s.v) { snprintf(b, sizeof b, "%s %6d %6d\n", l.seq.c_str(), l.e37, l.dH); o += b; }
    }
    o += "\n# END\n";
    return o;
}

// ---------------------------------------------------------------- temperature rescaling -> device tables

inline int rescale(int g37, int dH, double tempf)
{
    // ViennaRNA RESCALE_dG: evaluated in double, stored into an int (truncation toward zero)
    return (int)((double)dH - ((double)dH - (double)g37) * tempf);
}

inline uint32_t loop_key_host(const char *s, int m)
{
    uint32_t k = 0;
    for (int t = 0; t < m; t++) {
        const int c = s[t] == 'A' ? 1 : s[t] == 'C' ? 2 : s[t] == 'G' ? 3 : s[t] == 'U' ? 4 : 0;
        k |= (uint32_t)c << (3 * t);
    }
    return k;
}

// Fills the device-layout tables for `temp` (degrees C).  temp != 37 needs the enthalpies of a loaded file.
inline bool scaled_tables(const ParamSet &P, double temp, EnergyTables *h, std::string &err)
{
    const bool at37 = (temp == 37.0);
    if (!at37 && !P.has_dH) {
        err = "temp != 37 needs the enthalpy tables of a ViennaRNA parameter file (rafft_load_params); the built-in tables are 37 C only";
        return false;
    }
    const double tempf = (temp + K0) / TMEASURE;
    auto sc = [&](int g, int dh) { return at37 ? g : (g >= INF_ ? INF_ : rescale(g, dh, tempf)); };
    auto neg = [&](int v) { return v > 0 ? 0 : v; };       // dangles=2: positive dangle / multi / exterior mismatch terms are dropped
    auto fits = [&](int &v) { if (v > 30000) v = 30000; return v >= -32768; };   // INF entries of 16-bit tables: +300 kcal/mol forbids as well
    memset(h, 0, sizeof *h);
    bool ok = true;
    for (int a = 1; a < 7; a++) for (int b = 1; b < 7; b++) { int v = sc(P.stack[0][a][b], P.stack[1][a][b]); ok &= fits(v); h->s.stack[a][b] = (int16_t)v; }
    for (int i = 0; i < 31; i++) {
        h->s.hairpin[i] = sc(P.hairpin[0][i], P.hairpin[1][i]);
        h->s.bulge[i] = sc(P.bulge[0][i], P.bulge[1][i]);
        h->s.interior[i] = sc(P.interior[0][i], P.interior[1][i]);
    }
    for (int t = 1; t < 7; t++) for (int a = 0; a < 5; a++) {
        int v = neg(sc(P.d5[0][t][a], P.d5[1][t][a])); ok &= fits(v); h->s.d5[t][a] = (int16_t)v;
        v = neg(sc(P.d3[0][t][a], P.d3[1][t][a])); ok &= fits(v); h->s.d3[t][a] = (int16_t)v;
        for (int b = 0; b < 5; b++) {
            v = sc(P.mmH[0][t][a][b], P.mmH[1][t][a][b]); ok &= fits(v); h->s.mmH[t][a][b] = (int16_t)v;
            v = sc(P.mmI[0][t][a][b], P.mmI[1][t][a][b]); ok &= fits(v); h->s.mmI[t][a][b] = (int16_t)v;
            v = sc(P.mm1n[0][t][a][b], P.mm1n[1][t][a][b]); ok &= fits(v); h->s.mm1n[t][a][b] = (int16_t)v;
            v = sc(P.mm23[0][t][a][b], P.mm23[1][t][a][b]); ok &= fits(v); h->s.mm23[t][a][b] = (int16_t)v;
            v = neg(sc(P.mmM[0][t][a][b], P.mmM[1][t][a][b])); ok &= fits(v); h->s.mmM[t][a][b] = (int16_t)v;
            v = neg(sc(P.mmE[0][t][a][b], P.mmE[1][t][a][b])); ok &= fits(v); h->s.mmE[t][a][b] = (int16_t)v;
        }
    }
    for (int t = 1; t < 7; t++) for (int u = 1; u < 7; u++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
        int v = sc(P.int11[0][t][u][a][b], P.int11[1][t][u][a][b]); ok &= fits(v); h->b.int11[t][u][a][b] = (int16_t)v;
        for (int c = 0; c < 5; c++) {
            v = sc(P.int21[0][t][u][a][b][c], P.int21[1][t][u][a][b][c]); ok &= fits(v); h->b.int21[t][u][a][b][c] = (int16_t)v;
            for (int d = 0; d < 5; d++) {
                v = sc(P.int22[0][t][u][a][b][c][d], P.int22[1][t][u][a][b][c][d]); ok &= fits(v); h->b.int22[t][u][a][b][c][d] = (int16_t)v;
            }
        }
    }
    if (!ok) { err = "a table value does not fit 16 bits"; return false; }
    // The built-in set at 37 C: bit 0 of an interior-loop table entry says "rule / model value: no reference-held energy row
    // exercises it" (SmallT::lsb; the values are multiples of 10 dcal, the device strips the bit) - a fold can then say how many of
    // its stem energies involved one.  A loaded file's entries are all ViennaRNA's: no marks.
    h->s.lsb = 0;
    if (P.builtin_set && at37) {
        bool even = true;
        for (int t = 1; t < 7 && even; t++) for (int u = 1; u < 7; u++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
            even &= !(h->b.int11[t][u][a][b] & 1);
            for (int c = 0; c < 5; c++) { even &= !(h->b.int21[t][u][a][b][c] & 1); for (int d = 0; d < 5; d++) even &= !(h->b.int22[t][u][a][b][c][d] & 1); }
        }
        for (int t = 1; t < 7; t++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) even &= !((h->s.mmI[t][a][b] | h->s.mm1n[t][a][b] | h->s.mm23[t][a][b]) & 1);
        for (int i = 0; i < 31; i++) even &= !((h->s.bulge[i] | h->s.interior[i]) & 1);
        if (even) {
            h->s.lsb = 1;
            for (int t = 1; t < 7; t++) for (int u = 1; u < 7; u++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
                h->b.int11[t][u][a][b] |= (int16_t)!t04_int11_pinned[t][u][a][b];
                for (int c = 0; c < 5; c++) {
                    h->b.int21[t][u][a][b][c] |= (int16_t)!t04_int21_pinned[t][u][a][b][c];
                    for (int d = 0; d < 5; d++) h->b.int22[t][u][a][b][c][d] |= (int16_t)!t04_int22_pinned[t][u][a][b][c][d];
                }
            }
            for (int t = 1; t < 7; t++) for (int a = 0; a < 5; a++) for (int b = 0; b < 5; b++) {
                h->s.mmI[t][a][b] |= (int16_t)!t04_mismatch_interior_pinned[t][a][b];
                h->s.mm1n[t][a][b] |= (int16_t)!t04_mismatch_interior_1n_pinned[t][a][b];
                h->s.mm23[t][a][b] |= (int16_t)!t04_mismatch_interior_23_pinned[t][a][b];
            }
            for (int i = 1; i < 31; i++) {
                if (h->s.bulge[i] < INF_) h->s.bulge[i] |= !t04_bulge_pinned[i];
                if (h->s.interior[i] < INF_) h->s.interior[i] |= !t04_interior_pinned[i];
            }
        }
    }
    h->s.ml_base = sc(P.ml_base[0], P.ml_base[1]); h->s.ml_closing = sc(P.ml_closing[0], P.ml_closing[1]);
    h->s.ml_intern = sc(P.ml_intern[0], P.ml_intern[1]);
    h->s.ninio = sc(P.ninio[0], P.ninio[1]); h->s.max_ninio = P.max_ninio;
    h->s.term_au = sc(P.term_au[0], P.term_au[1]);
    const double lxc = at37 ? P.lxc : P.lxc * tempf;
    for (int sz = 31; sz <= RAFFT_MAX_LEN + 1; sz++) h->b.logext[sz] = (int)(lxc * log(sz / 30.));
    auto put = [&](uint32_t key, int size, int e) {
        const uint32_t k = key | sp_tag(size);
        uint32_t sl = sp_slot(k);
        while (h->s.sp_key[sl]) { if (h->s.sp_key[sl] == k) return; sl = (sl + 1) & 127u; }    // a sequence listed twice: the first entry wins (strstr)
        h->s.sp_key[sl] = k; h->s.sp_e[sl] = e;
        // (the filter in front of the table: closing pair and the loop's two end bases, e_hairpin)
        auto base = [&](int t) { return (int)((key >> (3 * t)) & 7u); };
        const uint32_t fi = sp_filter_index(size, base(0), base(1), base(size), base(size + 1));
        h->s.sp_filter[fi >> 5] |= 1u << (fi & 31u);
    };
    // stacking energies by the packed bases of a contiguous stem (SmallT::stk4, stem_stack_packed)
    for (int i = 0; i < 256; i++) {
        const int x5t = (i & 3) + 1, x5p = ((i >> 2) & 3) + 1, x3p = ((i >> 4) & 3) + 1, x3t = ((i >> 6) & 3) + 1;
        const int ty = pair_type(x5t, x3t), ty_in = pair_type(x5p, x3p);
        h->s.stk4[i] = (ty && ty_in) ? h->s.stack[ty][rtype(ty_in)] : 0;
    }
    for (const SpecialLoop &l : P.tri) put(loop_key_host(l.seq.c_str(), 5), 3, sc(l.e37, l.dH));
    for (const SpecialLoop &l : P.tetra) put(loop_key_host(l.seq.c_str(), 6), 4, sc(l.e37, l.dH));
    for (const SpecialLoop &l : P.hexa) put(loop_key_host(l.seq.c_str(), 8), 6, sc(l.e37, l.dH));
    return true;
}

} // namespace rafft_par
