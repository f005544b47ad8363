// zeldovich_main.cpp — `zeldovich <param_file>`: the drop-in command line of the reference
// (src/zeldovich.cpp:848-1032) on top of the MI355X library.  Host C++ only; everything heavy is
// behind the C ABI of include/zeldovich_hip.h.
//
// Same external surface: one argument, same parameter keys, ic_{z*CPD/PPD} files of ICFormat records
// under InitialConditionsDirectory, optional density file, progress on stderr, exit code 0 / 1.
// Differences that are deliberate and documented in DESIGN.md: no FFTW wisdom file, no block files
// (DISK mode is replaced by HBM residency + z-residue streaming), planes may be produced out of z
// order and are therefore placed with pwrite at their final offset instead of being appended.
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <map>
#include <string>

#include "../../include/zeldovich_hip.h"

namespace fs = std::filesystem;

struct Writer {
    zd_params p;
    fs::path dir;
    int recsize = 0;
    int64_t plane_bytes = 0;
    std::map<int, int> fds;  // ic file index -> fd
    int dens_fd = -1;
    size_t bytes_written = 0;
    double seconds = 0;

    // first z stored in file `f`: smallest z with z*cpd/ppd == f   (output.cpp:208)
    int64_t first_z_of_file(int f) const {
        if (p.cpd <= 0) return 0;  // CPD <= 0: z*cpd/ppd == 0 for every plane, everything lands in ic_0 (as in the reference)
        int64_t z = ((int64_t) f * p.ppd + p.cpd - 1) / p.cpd;
        while (z > 0 && (z - 1) * p.cpd / p.ppd == f) z--;
        while (z * p.cpd / p.ppd < f) z++;
        return z;
    }
    int fd_for(int f) {
        auto it = fds.find(f);
        if (it != fds.end()) return it->second;
        const fs::path fn = dir / ("ic_" + std::to_string(f));
        // planes are placed with pwrite, so a file is opened once per run: truncate whatever an earlier (longer) run left
        int fd = open(fn.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fd < 0) {
            fprintf(stderr, "Could not open output file \"%s\"\n", fn.c_str());
            exit(1);
        }
        fds[f] = fd;
        return fd;
    }
    static int callback(void *user, int64_t z, int64_t n, const void *records, const float *density) {
        Writer *w = (Writer *) user;
        const auto t0 = std::chrono::steady_clock::now();
        if (records) {
            const int f = (int) (z * w->p.cpd / w->p.ppd);
            // with ZD_qoneslab only one plane is written; the reference appends it at offset 0
            const int64_t zrel = w->p.qoneslab >= 0 ? 0 : z - w->first_z_of_file(f);
            const int fd       = w->fd_for(f);
            const size_t nb    = (size_t) n * w->recsize;
            if (pwrite(fd, records, nb, (off_t) (zrel * w->plane_bytes)) != (ssize_t) nb) {
                fprintf(stderr, "Short write on ic_%d\n", f);
                return 1;
            }
            w->bytes_written += nb;
        }
        if (density && w->dens_fd >= 0) {
            const size_t nb    = (size_t) n * sizeof(float);
            const int64_t zrel = w->p.qoneslab >= 0 ? 0 : z;
            if (pwrite(w->dens_fd, density, nb, (off_t) (zrel * (int64_t) nb)) != (ssize_t) nb) return 1;
            w->bytes_written += nb;
        }
        w->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return 0;
    }
    void close_all() {
        for (auto &kv : fds) close(kv.second);
        if (dens_fd >= 0) close(dens_fd);
    }
};

// SetupOutputDir: src/output.cpp:236-251
static void setup_output_dir(const fs::path &dir) {
    if (fs::exists(dir)) {
        for (const auto &entry : fs::directory_iterator(dir)) {
            if (entry.is_regular_file()) {
                const std::string fn = entry.path().filename().string();
                if (fn.compare(0, 3, "ic_") == 0 || fn.compare(0, 10, "zeldovich.") == 0) fs::remove(entry.path());
            }
        }
    }
    fs::create_directories(dir);
}

static double cube(double a) { return a == 0.0 ? 0.0 : a * a * a; }

int main(int argc, char *argv[]) {
    if (argc != 2) {
        fprintf(stderr, "Usage: %s param_file\n", argv[0]);
        exit(1);
    }
    const auto t_start = std::chrono::steady_clock::now();
    auto elapsed = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };

    zd_params p;
    zd_param_strings s;
    if (zd_params_from_file(argv[1], &p, &s)) {
        printf("Invalid Parameters given \n");
        exit(1);
    }
    zd_pk pk;
    zd_pk_handle *pkh = nullptr;
    if (s.Pk_filename[0]) {
        if (zd_pk_create_from_file(s.Pk_filename, s.Pk_scale, s.Pk_norm, s.Pk_sigma, s.Pk_sigma_ratio, s.Pk_smooth,
                                   s.qPk_fix_to_mean, p.boxsize, &pkh, &pk))
            return 1;
    } else {
        if (zd_pk_create_powerlaw(s.Pk_powerlaw_index, s.Pk_norm, s.Pk_sigma, s.Pk_sigma_ratio, s.Pk_smooth,
                                  s.qPk_fix_to_mean, p.boxsize, &pkh, &pk))
            return 1;
    }
    const int narray   = p.qdensity == 2 ? 1 : (p.qPLT ? 4 : 2);  // zeldovich.cpp:871-876
    const double memory = cube(p.ppd / 1024.0) * narray * 16.0;

    Writer w;
    w.p   = p;
    w.dir = s.output_dir;
    setup_output_dir(w.dir);
    static const int recsizes[4] = {32, 32, 56, 12};
    w.recsize     = p.qdensity == 2 ? 0 : recsizes[p.icformat];
    w.plane_bytes = (int64_t) p.ppd * p.ppd * w.recsize;
    if (p.qdensity) {  // InitOutputBuffers: output.cpp:282-288; "density{:d}" is an fmt pattern on ppd
        std::string name = s.density_filename;
        const size_t pos = name.find("{:d}");
        if (pos != std::string::npos) name.replace(pos, 4, std::to_string((long long) p.ppd));
        const fs::path path = w.dir / name;
        w.dens_fd           = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (w.dens_fd < 0) {
            fprintf(stderr, "Could not open density file \"%s\"\n", path.c_str());
            return 1;
        }
    }
    fprintf(stderr, "MI355X build: the whole problem (or one z-residue pass of it) resides in HBM.\n");
    fprintf(stderr, "Total array size: %5.3f GiB\n", memory);

    double *eig = nullptr;
    int64_t eig_ppd = 0;
    if (p.qPLT && zd_load_eigmodes(s.PLT_filename, &eig, &eig_ppd)) exit(1);
    if (p.k_cutoff != 1)
        fprintf(stderr, "Using k_cutoff = %f (effective ppd = %d)\n", p.k_cutoff, (int) (p.ppd / p.k_cutoff + .5));
    fprintf(stderr, "Preamble took %f seconds\n", elapsed());

    zd_stats st;
    memset(&st, 0, sizeof(st));
    if (zd_generate(&p, &pk, eig, eig_ppd, Writer::callback, &w, &st)) {
        fprintf(stderr, "zeldovich: generation failed\n");
        exit(1);
    }
    fprintf(stderr, "Grid -> displacements took %f seconds on the GPU (stream factor %d, modes %s)\n", st.seconds_total,
            st.stream_factor, st.modes_cached ? "cached in HBM" : "generated per pass");
    fprintf(stderr, "Time so far: %f seconds\n", elapsed());

    // zeldovich.cpp:987-1011
    fprintf(stderr, "The rms density variation of the pixels is %f\n", sqrt(st.density_variance / cube((double) p.ppd)));
    fprintf(stderr, "This could be compared to the P(k) prediction of %f\n",
            zd_pk_sigmaR(&pk, (p.boxsize / p.ppd) / 4.0) * pow(p.boxsize, 1.5));
    if (p.qdensity != 2) {
        fprintf(stderr, "The maximum component-wise displacements are (%g, %g, %g), same units as BoxSize.\n",
                st.max_disp[0], st.max_disp[1], st.max_disp[2]);
        fprintf(stderr,
                "For Abacus' 2LPT implementation to work (assuming FINISH_WAIT_RADIUS = 1),\n\tthis implies a maximum CPD of %d\n",
                (int) (p.boxsize / (2 * fabs(st.max_disp[2]))));
    }
    w.close_all();
    fprintf(stderr, "WriteParticlesSlab took %.3g sec to write %.3g MB ==> %.3g MB/sec\n", w.seconds,
            w.bytes_written / 1e6, w.bytes_written / 1e6 / (w.seconds > 0 ? w.seconds : 1e-9));
    if (eig) zd_free(eig);
    zd_pk_destroy(pkh);
    const double tot = elapsed();
    fprintf(stderr, "zeldovich took %.4g sec for ppd %lld ==> %.3g Mpart/sec\n", tot, (long long) p.ppd,
            (double) s.np / 1e6 / tot);
    return 0;
}
