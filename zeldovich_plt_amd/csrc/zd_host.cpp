// zd_host.cpp — host-side setup that stays on the CPU, behind the C ABI of include/zeldovich_hip.h:
//   * the parameter file reader + Parameters::setup      (src/parameters.cpp:11-197; the reference
//     parses with the flex/bison ParseHeader library — this is a hand-written reader of the same grammar:
//     comments and ## blocks, `include`, continuation lines, quoted strings, ints, floats with Fortran D
//     exponents, logical keywords, vectors, `vcounter` / `vector` blocks, `mapvar` aliases)
//   * PowerSpectrum tables and normalisation             (src/power_spectrum.cpp:50-223,
//     include/spline_function.h:77-163)
//   * the PLT eigenmode file loader                      (src/zeldovich.cpp:794-830)
// No GPU code here; the amplitudes produced on the device depend on these numbers to ~1e-16, so
// the arithmetic order of the reference is kept.
#include <algorithm>
#include <cassert>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/zeldovich_hip.h"

namespace {

// ---------------------------------------------------------------------------------------------
// P(k) table: natural cubic spline through (ln k, ln P)  (what SplineFunction does for the reference,
// include/spline_function.h:77-163).  Every amplitude the GPU produces carries these numbers, so each floating-point
// operation below happens in the reference's order; the code around them is this library's own.

struct Spline {
    std::vector<double> x, y, y2;

    // The reference orders its table with a diminishing-increment sort whose comparison looks ONE SLOT TO THE RIGHT of
    // the element it then moves (spline_function.h:92: the key array is addressed 0-based there, the moved arrays
    // 1-based).  An ascending table — every P(k) file in practice — passes through unchanged; anything else comes out
    // in the reference's order, not necessarily sorted.  Same increments (1, 4, 13, ... largest first), same
    // comparisons, written 0-based.
    void order_like_reference() {
        const long n = (long) x.size();
        std::vector<long> gaps;
        for (long g = 1; g <= n; g = 3 * g + 1) gaps.push_back(g);
        if (gaps.empty()) return;
        // the reference starts from the first increment ABOVE n divided by 3, i.e. the largest one <= n ... 1
        for (auto it = gaps.rbegin(); it != gaps.rend(); ++it) {
            const long gap = *it;
            for (long p = gap; p < n; p++) {
                const double kx = x[p], ky = y[p];
                long q = p;
                while (x[q + 1 - gap] > kx) {  // (the neighbour to the right of slot q - gap)
                    x[q] = x[q - gap];
                    y[q] = y[q - gap];
                    q -= gap;
                    if (q < gap) break;
                }
                x[q] = kx;
                y[q] = ky;
            }
        }
    }

    // second derivatives of the natural spline: forward elimination of the tridiagonal system, then back substitution
    void build() {
        const long n = (long) x.size();
        order_like_reference();
        y2.assign(n, 0.0);
        std::vector<double> rhs(n, 0.0), slope(n, 0.0);
        for (long i = 0; i + 1 < n; i++) slope[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]);
        for (long i = 1; i + 1 < n; i++) {
            const double span  = x[i + 1] - x[i - 1];
            const double ratio = (x[i] - x[i - 1]) / span;
            const double pivot = ratio * y2[i - 1] + 2.0;
            y2[i]  = (ratio - 1.0) / pivot;
            rhs[i] = (6.0 * (slope[i] - slope[i - 1]) / span - ratio * rhs[i - 1]) / pivot;
        }
        y2[n - 1] = 0.0;  // natural end (the reference's (un - qn u)/(qn y2 + 1) with un = qn = 0)
        for (long i = n - 2; i >= 0; i--) y2[i] = y2[i] * y2[i + 1] + rhs[i];
    }
};

// value of the spline at v; outside the table the first / last cubic piece continues (the bisection of
// spline_function.h:141-163 ends on the last node <= v, kept inside [0, n-2])
double spline_val(int n, const double *x, const double *y, const double *y2, double v) {
    long lo = 0, hi = n - 1;  // halving by hand, not std::upper_bound: on a table the sort above left unordered the two differ
    while (hi - lo > 1) {
        const long mid = lo + (hi - lo) / 2;
        (x[mid] > v ? hi : lo) = mid;
    }
    const double width = x[hi] - x[lo];
    const double wl = (x[hi] - v) / width, wh = (v - x[lo]) / width;
    return wl * y[lo] + wh * y[hi] + ((wl * wl * wl - wl) * y2[lo] + (wh * wh * wh - wh) * y2[hi]) * (width * width) / 6.0;
}

double pk_power(const zd_pk *pk, double k) {  // power_spectrum.cpp:225-261
    if (k <= 0.0) return 0.0;
    if (pk->is_powerlaw) return std::pow(k, pk->powerlaw_index) * std::exp(-k * k * pk->Pk_smooth2) * pk->normalization;
    static bool already_warned = false;
    if (k > pk->kmax && !already_warned) {
        fprintf(stderr,
                "\n*** WARNING: power spectrum spline interpolation was requested\n"
                "past the maximum k (%f) that was provided in the input power\n"
                "spectrum file.  The extrapolation should be well-behaved, but\n"
                "make sure that this was expected.  Provide a power spectrum\n"
                "that goes to at least k=10 (or higher if your k_Nyquist demands\n"
                "it) to get rid of this warning.\n\n",
                pk->kmax);
        already_warned = true;
    }
    return std::exp(spline_val(pk->n, pk->x, pk->y, pk->y2, std::log(k)) - k * k * pk->Pk_smooth2) * pk->normalization;
}

// sigma_R^2 = int dk k^2 P(k) W^2(kR) / (2 pi^2), W = the top-hat window (series below kR = 1e-3); integrand of
// power_spectrum.cpp:50-58
struct TopHatVariance {
    const zd_pk *pk;
    double R;
    double operator()(double k) const {
        const double kr = k * R;
        const double window = kr <= 1e-3 ? 1 - kr * kr / 10.0 : 3.0 * (std::sin(kr) - kr * std::cos(kr)) / kr / kr / kr;
        return 0.5 / M_PI / M_PI * k * k * window * window * pk_power(pk, k);
    }
};

// Romberg integration as the reference runs it (power_spectrum.cpp:93-128): trapezoid refined by midpoints taken in
// ascending order, Richardson columns with factors 4, 16, ...; stops when the diagonal moves by less than `tol`
// relatively (at least two refinements, at most 32).  Only the previous row of the tableau is needed.
template <class F>
double romberg(const F &f, double lo, double hi, double tol, double *achieved) {
    constexpr int kMaxLevels = 32;
    std::vector<double> prev(1, 0.0), cur;
    double half = 0.5 * (hi - lo);
    prev[0]     = half * (f(lo) + f(hi));
    double diag_before = prev[0], diag = prev[0];
    for (int level = 1; level <= kMaxLevels; level++) {
        double mid = 0;
        for (uint64_t m = 1; m <= (1ULL << (level - 1)); m++) mid += f(lo + (2 * m - 1) * half);
        cur.assign(level + 1, 0.0);
        cur[0]       = 0.5 * prev[0] + half * mid;
        double power = 1;
        for (int c = 1; c < level; c++) {
            power *= 4;
            cur[c] = cur[c - 1] + (cur[c - 1] - prev[c - 1]) / (power - 1);
        }
        half *= 0.5;
        diag_before = prev[level - 1 > 0 ? level - 2 : 0];
        diag        = cur[level - 1];
        prev.swap(cur);
        if (level > 1 && std::fabs(diag - diag_before) < tol * std::fabs(diag)) break;
    }
    *achieved = (diag - diag_before) / diag;
    return diag;
}

double pk_sigmaR(const zd_pk *pk, double R) {  // power_spectrum.cpp:60-89
    if (!pk->is_powerlaw) {
        const double want = 1e-6;
        double got        = 1.0;
        const double sig  = std::sqrt(romberg(TopHatVariance{pk, R}, 0, 10.0, want, &got));
        if (got > want) {
            fprintf(stderr,
                    "Error: actual Romberg integration precision (%g) is greater than the target precision (%g); halting.\n",
                    got, want);
            exit(1);
        }
        return sig;
    }
    // power law: closed form of the same integral
    const double idx = pk->powerlaw_index;
    const double var = 9 * std::pow(R, -idx - 3) / (2 * M_PI * std::sqrt(M_PI)) * std::tgamma((3 + idx) / 2.)
                       / (std::tgamma((2 - idx) / 2.) * (idx - 3) * (idx - 1));
    return std::sqrt(var * pk->normalization);
}

}  // namespace

struct zd_pk_handle {
    Spline sp;
};

static void normalize(zd_pk *pk, double Pk_norm, double Pk_sigma, double Pk_sigma_ratio, double Pk_smooth,
                      int fix_to_mean, double boxsize) {  // power_spectrum.cpp:186-223
    pk->Pk_smooth2    = 0.0;
    pk->normalization = 1.0;
    if (Pk_norm > 0.0) {
        fprintf(stderr, "Input sigma(%f) = %.6g\n", Pk_norm, pk_sigmaR(pk, Pk_norm));
        if (Pk_sigma > 0) {
            pk->normalization = Pk_sigma / pk_sigmaR(pk, Pk_norm);
            pk->normalization *= pk->normalization;
        } else if (Pk_sigma_ratio > 0) {
            pk->normalization = Pk_sigma_ratio * Pk_sigma_ratio;
        } else {
            assert(Pk_sigma > 0 || Pk_sigma_ratio > 0);
        }
        fprintf(stderr, "Final sigma(%f) = %.6g\n", Pk_norm, pk_sigmaR(pk, Pk_norm));
    }
    pk->normalization /= boxsize * boxsize * boxsize;
    pk->Pk_smooth2  = Pk_smooth * Pk_smooth;
    pk->fixed_power = fix_to_mean;
    if (pk->fixed_power) fprintf(stderr, "Fixing density mode amplitudes to sqrt(P(k))\n");
}

extern "C" {

int zd_pk_create_from_file(const char *path, double Pk_scale, double Pk_norm, double Pk_sigma,
                           double Pk_sigma_ratio, double Pk_smooth, int fix_to_mean, double boxsize,
                           zd_pk_handle **hout, zd_pk *pk) {  // power_spectrum.cpp:130-171
    fprintf(stderr, "Loading power spectrum from file \"%s\"\n", path);
    FILE *fp = fopen(path, "r");
    if (fp == NULL) {
        fprintf(stderr, "Power spectrum file \"%s\" not found; exiting.\n", path);
        return 1;
    }
    zd_pk_handle *h = new zd_pk_handle;
    double kmin = std::numeric_limits<double>::max(), kmax = std::numeric_limits<double>::min();
    char line[200];
    double k = 0, P = 0;
    while (fgets(line, 200, fp) != NULL) {
        if (line[0] == '#') continue;
        sscanf(line, "%lf %lf", &k, &P);
        if (k < 0.0) continue;
        if (P < 0.0) continue;
        double ks = k * Pk_scale;
        if (ks > 0.0) {
            h->sp.x.push_back(std::log(ks));
            h->sp.y.push_back(std::log(P));
            kmin = std::min(ks, kmin);
        } else {
            h->sp.x.push_back(-1e3);
            h->sp.y.push_back(std::log(P));
        }
        kmax = std::max(ks, kmax);
    }
    fclose(fp);
    if (h->sp.x.size() < 3) {
        fprintf(stderr, "Power spectrum file \"%s\" has fewer than 3 usable rows.\n", path);
        delete h;
        return 1;
    }
    h->sp.build();
    memset(pk, 0, sizeof(*pk));
    pk->n  = (int) h->sp.x.size();
    pk->x  = h->sp.x.data();
    pk->y  = h->sp.y.data();
    pk->y2 = h->sp.y2.data();
    pk->kmax           = kmax;
    pk->kmin           = kmin;
    pk->powerlaw_index = 1000;
    normalize(pk, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth, fix_to_mean, boxsize);
    *hout = h;
    return 0;
}

int zd_pk_create_powerlaw(double index, double Pk_norm, double Pk_sigma, double Pk_sigma_ratio, double Pk_smooth,
                          int fix_to_mean, double boxsize, zd_pk_handle **hout, zd_pk *pk) {  // :173-184
    if (index == 1000) return 1;
    fprintf(stderr, "Initializing power spectrum with power law index %g\n", index);
    zd_pk_handle *h = new zd_pk_handle;
    memset(pk, 0, sizeof(*pk));
    pk->powerlaw_index = index;
    pk->is_powerlaw    = 1;
    pk->kmax           = std::numeric_limits<double>::min();
    pk->kmin           = 1e-4;  // Arbitrary; used by f_NL (power_spectrum.cpp:180)
    normalize(pk, Pk_norm, Pk_sigma, Pk_sigma_ratio, Pk_smooth, fix_to_mean, boxsize);
    *hout = h;
    return 0;
}

double zd_pk_power(const zd_pk *pk, double k) { return pk_power(pk, k); }
double zd_pk_sigmaR(const zd_pk *pk, double R) { return pk_sigmaR(pk, R); }
void zd_pk_destroy(zd_pk_handle *h) { delete h; }

// load_eigmodes: int32 ppd + ppd*ppd*(ppd/2+1)*4 doubles  (zeldovich.cpp:794-830)
int zd_load_eigmodes(const char *path, double **eig, int64_t *eig_ppd) {
    fprintf(stderr, "Using PLT eigenmodes.\n");
    std::ifstream f(path, std::ios::in | std::ios::binary | std::ios::ate);
    if (!f) {
        fprintf(stderr, "[Error] Could not open eigenmode file \"%s\".\n", path);
        return 1;
    }
    const std::streampos size = f.tellg();
    f.seekg(0, std::ios::beg);
    int32_t ppd32 = 0;
    f.read((char *) &ppd32, sizeof(ppd32));
    const int64_t ep    = ppd32;
    const size_t nelem  = (size_t) ep * ep * (ep / 2 + 1) * 4;
    const size_t nbytes = nelem * sizeof(double);
    if ((size_t) size != nbytes + sizeof(ppd32)) {
        fprintf(stderr, "[Error] Eigenmode file \"%s\" of size %lld did not match expected size %zu from eig_vecs_ppd %lld.\n",
                path, (long long) size, nbytes, (long long) ep);
        return 1;
    }
    double *buf = NULL;
    if (posix_memalign((void **) &buf, 4096, nbytes) != 0) return 1;
    f.read((char *) buf, nbytes);
    *eig     = buf;
    *eig_ppd = ep;
    return 0;
}
void zd_free(void *p) { free(p); }

// ---------------------------------------------------------------------------------------------
// parameter file

static std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char) s[a])) a++;
    while (b > a && isspace((unsigned char) s[b - 1])) b--;
    return s.substr(a, b - a);
}
static std::string unquote(const std::string &s) {
    std::string t = trim(s);
    if (t.size() >= 2 && ((t.front() == '"' && t.back() == '"') || (t.front() == '\'' && t.back() == '\'')))
        return t.substr(1, t.size() - 2);
    return t;
}
static double to_double(std::string s) {
    for (auto &c : s)
        if (c == 'D' || c == 'd') c = 'e';  // Fortran exponents (phScanner.ll)
    return strtod(s.c_str(), NULL);
}

// split a statement into tokens the way phScanner.ll does: blanks separate, quoted strings are one token
static std::vector<std::string> tokens_of(const std::string &line) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < line.size()) {
        if (isspace((unsigned char) line[i])) {
            i++;
            continue;
        }
        size_t j = i;
        if (line[i] == '"' || line[i] == '\'') {
            const char q = line[i];
            j = line.find(q, i + 1);
            j = j == std::string::npos ? line.size() : j + 1;
        } else {
            while (j < line.size() && !isspace((unsigned char) line[j])) j++;
        }
        out.push_back(line.substr(i, j - i));
        i = j;
    }
    return out;
}

struct HeaderState {
    std::map<std::string, std::string> kv;
    std::map<std::string, std::string> alias;  // mapvar: variable -> the variable whose storage it shares
    std::string vcounter;                      // name of the current element counter (vcounter statement)
    std::vector<std::string> vec_ids;          // open `vector` block: the variables its rows fill
    std::vector<std::vector<std::string>> vec_cols;
    bool in_vector = false;
    void assign(const std::string &key, const std::string &val) {
        auto it = alias.find(key);
        kv[it == alias.end() ? key : it->second] = val;
    }
    // a vector block ends at the first statement that is not a row of values (phParser.yy:117-128): every variable gets
    // its column (values separated by blanks, which is how the installed vectors are written with `=`), the counter the
    // number of rows
    void close_vector() {
        if (!in_vector) return;
        for (size_t i = 0; i < vec_ids.size(); i++) {
            std::string v;
            for (const auto &e : vec_cols[i]) v += (v.empty() ? "" : " ") + e;
            assign(vec_ids[i], v);
        }
        if (!vcounter.empty()) kv[vcounter] = std::to_string(vec_cols.empty() ? 0 : vec_cols[0].size());
        in_vector = false;
        vec_ids.clear();
        vec_cols.clear();
    }
};

// Reads the statements of a parameter file the way the reference's ParseHeader does (scanner
// subprojects/ParseHeader/src/phScanner.ll:95-240, grammar phParser.yy:74-166): `#` comments, `##` block comments,
// quoted strings, a trailing backslash continues the statement on the next line, `include "file"` splices another file
// (depth <= 10, phScanner.ll:176-203), a ^B character ends the header;
//   id = value ...            assignment
//   vcounter id               id counts the rows of the vector blocks that follow (phDriver.cc:469-485)
//   vector id ...             the following lines of bare values are rows: column i fills variable i (:504-531)
//   mapvar base id ...        the ids share base's storage: assigning one assigns base (:433-467)
// anything else without `=` is the reference's syntax error "expecting '='".
static int read_statements(const char *path, HeaderState &H, int depth) {
    if (depth > 10) {
        fprintf(stderr, "ERROR: exceeded maximum include stack depth: 10.\n");
        return 1;
    }
    std::ifstream in(path);
    if (!in) {
        fprintf(stderr, depth ? "failed to open include file \"%s\". exiting...\n" : "Could not open parameter file \"%s\"\n", path);
        return 1;
    }
    std::string line, pending;
    bool in_block_comment = false;
    int lineno = 0;
    while (std::getline(in, line)) {
        lineno++;
        if (!line.empty() && line[0] == '\x02') break;  // ^B ends the header (ParseHeader convention)
        if (trim(line).rfind("##", 0) == 0 && line.rfind("##", 0) == 0) {  // `##` at line start toggles a block comment
            in_block_comment = !in_block_comment;
            continue;
        }
        if (in_block_comment) continue;
        // strip `#` comments outside quotes
        bool q = false;
        size_t cut = std::string::npos;
        for (size_t i = 0; i < line.size(); i++) {
            if (line[i] == '"') q = !q;
            if (line[i] == '#' && !q) {
                cut = i;
                break;
            }
        }
        if (cut != std::string::npos) line = line.substr(0, cut);
        // continuation: backslash + optional blanks at end of line
        std::string t = line;
        while (!t.empty() && (t.back() == ' ' || t.back() == '\t' || t.back() == '\r')) t.pop_back();
        if (!t.empty() && t.back() == '\\') {
            t.pop_back();
            pending += t + " ";
            continue;
        }
        line = pending + line;
        pending.clear();
        const std::string st = trim(line);
        if (st.empty()) {  // a bare end of statement closes an open vector block
            H.close_vector();
            continue;
        }
        if (st.compare(0, 7, "include") == 0 && st.size() > 7 && (isspace((unsigned char) st[7]) || st[7] == '"' || st[7] == '\'')) {
            H.close_vector();
            const std::string inc = unquote(st.substr(7));
            if (read_statements(inc.c_str(), H, depth + 1)) return 1;
            continue;
        }
        size_t eq = std::string::npos;  // first `=` outside quotes
        {
            char qc = 0;
            for (size_t i = 0; i < line.size(); i++) {
                if (qc) {
                    if (line[i] == qc) qc = 0;
                } else if (line[i] == '"' || line[i] == '\'')
                    qc = line[i];
                else if (line[i] == '=') {
                    eq = i;
                    break;
                }
            }
        }
        if (eq != std::string::npos) {
            H.close_vector();
            const std::string key = trim(line.substr(0, eq));
            if (key.empty()) continue;
            H.assign(key, trim(line.substr(eq + 1)));
            continue;
        }
        const std::vector<std::string> tok = tokens_of(st);
        if (tok[0] == "vcounter") {
            H.close_vector();
            if (tok.size() != 2) {
                fprintf(stderr, "%s:%d: ERROR: vcounter takes one variable name\n", path, lineno);
                return 1;
            }
            H.vcounter = tok[1];
            H.kv[tok[1]] = "0";
            continue;
        }
        if (tok[0] == "vector") {
            H.close_vector();
            if (H.vcounter.empty()) {
                fprintf(stderr, "%s:%d: ERROR: must specify a vcounter variable before vector statement.\n", path, lineno);
                return 1;
            }
            if (tok.size() < 2) {
                fprintf(stderr, "%s:%d: ERROR: vector statement without variables\n", path, lineno);
                return 1;
            }
            H.vec_ids.assign(tok.begin() + 1, tok.end());
            H.vec_cols.assign(H.vec_ids.size(), {});
            H.in_vector = true;
            continue;
        }
        if (tok[0] == "mapvar") {
            H.close_vector();
            if (tok.size() < 3) {
                fprintf(stderr, "%s:%d: ERROR: mapvar needs a base variable and at least one name\n", path, lineno);
                return 1;
            }
            for (size_t i = 2; i < tok.size(); i++) {
                auto it = H.alias.find(tok[i]);
                if (it != H.alias.end()) {
                    fprintf(stderr, "%s:%d: ERROR: variable \"%s\" already mapped to \"%s\"\n", path, lineno, tok[i].c_str(), it->second.c_str());
                    return 1;
                }
                H.alias[tok[i]] = tok[1];
            }
            continue;
        }
        if (H.in_vector) {  // one row: value i goes to variable i (extra values / short rows: phDriver's stuffit complains)
            if (tok.size() != H.vec_ids.size()) {
                fprintf(stderr, "%s:%d: ERROR: vector row has %zu values for %zu variables\n", path, lineno, tok.size(), H.vec_ids.size());
                return 1;
            }
            for (size_t i = 0; i < tok.size(); i++) H.vec_cols[i].push_back(tok[i]);
            continue;
        }
        fprintf(stderr, "%s:%d: ERROR: syntax error, unexpected %s \"%s\", expecting '%s='\n", path, lineno,
                (isalpha((unsigned char) tok[0][0]) || tok[0][0] == '_' || tok[0][0] == '.' || tok[0][0] == '$') ? "string" : "value",
                tok[0].c_str(), isalpha((unsigned char) tok[0][0]) ? "" : "identifier ");
        return 1;
    }
    H.close_vector();
    return 0;
}

int zd_params_from_file(const char *path, zd_params *p, zd_param_strings *s) {
    HeaderState H;
    if (read_statements(path, H, 0)) return 1;
    std::map<std::string, std::string> &kv = H.kv;

    // defaults: src/parameters.cpp:13-44
    memset(p, 0, sizeof(*p));
    memset(s, 0, sizeof(*s));
    p->numblock = 2;
    p->qoneslab = -1;
    p->f_cluster = 1;
    p->k_cutoff  = 1.;
    s->Pk_scale  = 1;
    s->Pk_powerlaw_index = 1000;
    strcpy(s->density_filename, "density{:d}");
    s->version = -1;
    s->n_s     = 1;
    s->Omega_M = 1.0;
    int cpd = 0;
    bool have_cpd = false;

    auto has = [&](const char *k) { return kv.find(k) != kv.end(); };
    auto must = [&](const char *k) {
        if (!has(k)) {
            fprintf(stderr, "Parameter \"%s\" must be defined in \"%s\"\n", k, path);
            return false;
        }
        return true;
    };
    // MUST_DEFINE keys: src/parameters.cpp:61-95
    const char *required[] = {"BoxSize", "ZD_Pk_scale", "NP", "ZD_NumBlock", "CPD", "ZD_Seed", "ZD_Pk_norm",
                              "ZD_Pk_smooth", "InitialConditionsDirectory", "InitialRedshift", "ICFormat"};
    for (const char *k : required)
        if (!must(k)) return 1;

    auto D = [&](const char *k, double &v) { if (has(k)) v = to_double(kv[k]); };
    auto I = [&](const char *k, int32_t &v) {
        if (!has(k)) return;
        std::string t = kv[k];
        for (auto &c : t) c = (char) tolower((unsigned char) c);
        if (t == "true" || t == ".true.")  // logical keywords of the ParseHeader grammar
            v = 1;
        else if (t == "false" || t == ".false.")
            v = 0;
        else
            v = (int32_t) strtol(kv[k].c_str(), NULL, 0);
    };
    auto S = [&](const char *k, char *dst, size_t cap) {
        if (has(k)) {
            std::string v = unquote(kv[k]);
            strncpy(dst, v.c_str(), cap - 1);
        }
    };
    D("BoxSize", p->boxsize);
    D("ZD_Pk_scale", s->Pk_scale);
    if (has("NP")) s->np = (int64_t) llround(to_double(kv["NP"]));
    I("ZD_NumBlock", p->numblock);
    if (has("CPD")) {
        cpd      = (int) strtol(kv["CPD"].c_str(), NULL, 0);
        have_cpd = true;
    }
    I("ZD_qdensity", p->qdensity);
    I("ZD_qoneslab", p->qoneslab);
    int32_t seed32 = 0;
    I("ZD_Seed", seed32);
    D("ZD_Pk_norm", s->Pk_norm);
    D("ZD_Pk_sigma", s->Pk_sigma);
    D("ZD_Pk_sigma_ratio", s->Pk_sigma_ratio);
    D("ZD_f_cluster", p->f_cluster);
    D("ZD_Pk_smooth", s->Pk_smooth);
    I("ZD_qPk_fix_to_mean", s->qPk_fix_to_mean);
    S("ZD_Pk_filename", s->Pk_filename, sizeof(s->Pk_filename));
    D("ZD_Pk_powerlaw_index", s->Pk_powerlaw_index);
    S("InitialConditionsDirectory", s->output_dir, sizeof(s->output_dir));
    S("ZD_density_filename", s->density_filename, sizeof(s->density_filename));
    D("InitialRedshift", p->z_initial);
    I("ZD_qonemode", p->qonemode);
    if (has("ZD_one_mode")) {
        std::stringstream ss(kv["ZD_one_mode"]);
        std::string tok;
        int i = 0;
        while (i < 3 && ss >> tok) {
            while (!tok.empty() && (tok.back() == ',')) tok.pop_back();
            p->one_mode[i++] = (int32_t) strtol(tok.c_str(), NULL, 0);
        }
    }
    I("ZD_qPLT", p->qPLT);
    S("ZD_PLT_filename", s->PLT_filename, sizeof(s->PLT_filename));
    I("ZD_qPLT_rescale", p->qPLTrescale);
    D("ZD_PLT_target_z", p->PLT_target_z);
    D("ZD_k_cutoff", p->k_cutoff);
    D("ZD_f_NL", s->f_NL);
    D("ZD_n_s", s->n_s);
    D("Omega_M", s->Omega_M);
    S("ICFormat", s->ICFormat, sizeof(s->ICFormat));
    I("ZD_Version", s->version);
    I("ZD_CornerModes", p->corner_modes);
    // optional MI355X knobs (not in the reference; reference .par files run unchanged)
    I("ZD_StreamFactor", p->stream_factor);
    I("ZD_NumGPU", p->ngpu);  // GPUs of this node to drive (default: 1)
    I("ZD_ExchangePlanes", p->exchange_planes);
    I("ZD_PassGroups", p->pass_groups);  // independent groups of GPUs, residue passes dealt round-robin (0: automatic)
    (void) have_cpd;
    p->cpd = cpd;

    // ---- Parameters::setup: src/parameters.cpp:97-197 ----
    if (s->version == -1) {
        fprintf(stderr,
                "\n*** ERROR: ZD_Version was not specified for zeldovich-PLT.  New ICs should\n"
                "    specify ZD_Version = 2; legacy ICs (pre-November 2019) should use\n"
                "    ZD_Version = 1 to reproduce the old phases.  Please specify one of\n"
                "    these in the parameter file.\n");
        return 1;
    }
    if (s->version != 1 && s->version != 2) {  // parameters.cpp:111
        fprintf(stderr, "Invalid Parameters given: assertion `version == 1 || version == 2' failed\n");
        return 1;
    }
    if (s->version == 1)  // parameters.cpp:113-120
        fprintf(stderr,
                "\n*** WARNING: ZD_Version = 1 selected: the phases of this legacy mode depend on ZD_NumBlock.\n"
                "    Use it only to reproduce old initial conditions; new ones should set ZD_Version = 2.\n\n");
    p->version = s->version;
    p->ppd = (int64_t) llround(cbrt((double) s->np));
    fprintf(stderr, "Generating ICs for ppd = %lld\n", (long long) p->ppd);
#define ZD_REQUIRE(cond)                                                             \
    if (!(cond)) {                                                                   \
        fprintf(stderr, "Invalid Parameters given: assertion `%s' failed\n", #cond); \
        return 1;                                                                    \
    }
    ZD_REQUIRE(p->ppd * p->ppd * p->ppd == s->np);
    ZD_REQUIRE(p->ppd <= ZD_MAX_PPD);
    if (s->version == 1 && p->k_cutoff != 1.) {  // keeps the streams aligned between different PPD (parameters.cpp:129-141)
        const int numblock_old = p->numblock;
        p->numblock = (int) (p->numblock * p->k_cutoff + .5);
        fprintf(stderr, "Note: using k_cutoff=%f means that we are using NumBlock=%d instead of the supplied value of NumBlock=%d\n",
                p->k_cutoff, p->numblock, numblock_old);
    }
    ZD_REQUIRE(!(p->boxsize <= 0.0));
    ZD_REQUIRE(!(p->ppd <= 0));
    ZD_REQUIRE(!(p->numblock <= 0));
    ZD_REQUIRE(!(s->Pk_scale <= 0.0));
    ZD_REQUIRE(!(s->Pk_norm < 0.0));
    if ((bool) (s->Pk_sigma > 0) == (bool) (s->Pk_sigma_ratio > 0)) {
        fprintf(stderr, "Must specify exactly one of Pk_sigma or Pk_sigma_ratio!\n");
        return 1;
    }
    ZD_REQUIRE(p->f_cluster > 0. && p->f_cluster <= 1.);
    ZD_REQUIRE((s->Pk_filename[0] != 0) != (bool) (s->Pk_powerlaw_index != 1000));
    if (s->Pk_powerlaw_index != 1000) ZD_REQUIRE(s->Pk_powerlaw_index <= 0);
    if (p->qPLT) ZD_REQUIRE(s->PLT_filename[0] != 0);
    ZD_REQUIRE(p->k_cutoff >= 1);
    if (p->qPLT) ZD_REQUIRE(strncmp(s->ICFormat, "RV", 2) == 0);
    p->f_NL    = s->f_NL;
    p->n_s     = s->n_s;
    p->Omega_M = s->Omega_M;
    if (s->f_NL != 0.) {  // parameters.cpp:181-194
        fprintf(stderr,
                "Generating local primordial non-Gaussianity, with parameters:\n - ZD_f_NL = %g\n - ZD_n_s = %g\n"
                " - Omega_M = %g\n - InitialRedshift = %g\n",
                s->f_NL, s->n_s, s->Omega_M, p->z_initial);
    }
    // block geometry asserts of BlockArray (src/block_array.cpp:38-40) — kept so that reference
    // parameter files are rejected in the same situations
    ZD_REQUIRE(p->ppd % 2 == 0);
    ZD_REQUIRE(p->numblock % 2 == 0);
    ZD_REQUIRE(p->ppd % p->numblock == 0);
#undef ZD_REQUIRE
    const double separation = p->boxsize / p->ppd;
    p->nyquist     = M_PI / separation;
    p->fundamental = 2.0 * M_PI / p->boxsize;
    p->seed        = (int64_t) seed32;  // int -> unsigned long sign-extends (power_spectrum.cpp:14)
    if (p->qonemode) fprintf(stderr, "one_mode: %d, %d, %d\n", p->one_mode[0], p->one_mode[1], p->one_mode[2]);

    if (!strcmp(s->ICFormat, "RVdoubleZel"))
        p->icformat = ZD_FMT_RVDOUBLEZEL;
    else if (!strcmp(s->ICFormat, "RVZel"))
        p->icformat = ZD_FMT_RVZEL;
    else if (!strcmp(s->ICFormat, "Zeldovich"))
        p->icformat = ZD_FMT_ZEL;
    else if (!strcmp(s->ICFormat, "ZelSimple"))
        p->icformat = ZD_FMT_ZELSIMPLE;
    else if (p->qdensity != 2) {
        fprintf(stderr, "Error: unknown ICFormat \"%s\". Aborting.\n", s->ICFormat);  // output.cpp:272-277
        return 1;
    }
    return 0;
}

}  // extern "C"
