// main.cpp -- headless command-line program with the reference executables' argument
// contracts (rendering is replaced by an optional binary dump):
//
//   BarnesHut <N> <Sim> <Iter>      src/gpu/BH/main.cu:86-117   (Sim 0 spiral 1 random 2 collision 3 solar)
//   DirectSum <N> <Sim> <Iter>      src/gpu/DS/directSum_kernel.cu:258-312 (Sim 0 spiral 1 random 2 solar)
//   nBodyCPU  <N> <Alg> <Sim>       src/cpu/main.cpp:79-122     (Alg 0 direct 1 Barnes-Hut; runs --iters steps)
//
// The program acts as the one it is invoked as (argv[0]); under any other name it is
// BarnesHut unless --alg direct is given.  Exactly three positional arguments, or none for
// the defaults (N 300, Sim 0, Iter 300), as in the reference; bad values print the reference's
// message and return -1.  Additive flags: --alg bh|direct, --seed S, --theta T, --dump FILE,
// --dump-every K, --ppm PREFIX (one image PREFIX_<step>.ppm per dumped step, drawn by the rule
// of the reference's storeFrame), --window W, --iters K (nBodyCPU form),
// --integrator euler|leapfrog, --energy (prints the conserved quantities before and after the
// run), --thickness T (three-dimensional run: the scenario's bodies get heights in [-T, T]; frames
// are then "NBDUMP3" records {x, y, z, vx, vy, vz, mass} and images are the x-y projection),
// --gpus G (one process per GPU: the program starts G ranks of itself, each advances a slice of
// the bodies; Barnes-Hut ranks exchange one record per cell of the level the tree closes at, or, with
// --exchange positions or where that does not apply, all positions by an RCCL all-gather every step;
// rank 0 reports), --quiet.
#include <hip/hip_runtime_api.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "nbody_sim.hpp"

namespace {

// One frame: "NBDUMP1\0", n, step, then n records {x, y, vx, vy, mass} of doubles.  A frame
// holds the full dynamical state, so it also serves as a checkpoint.
void dump_frame(std::FILE *f, const nb_body *b, int n, int step)
{
    const char magic[8] = {'N', 'B', 'D', 'U', 'M', 'P', '1', 0};
    const uint64_t hdr[2] = {(uint64_t)n, (uint64_t)step};
    std::fwrite(magic, 1, 8, f);
    std::fwrite(hdr, sizeof(uint64_t), 2, f);
    std::vector<double> rec(5 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        rec[5 * (size_t)i + 0] = b[i].position.x;
        rec[5 * (size_t)i + 1] = b[i].position.y;
        rec[5 * (size_t)i + 2] = b[i].velocity.x;
        rec[5 * (size_t)i + 3] = b[i].velocity.y;
        rec[5 * (size_t)i + 4] = b[i].mass;
    }
    std::fwrite(rec.data(), sizeof(double), rec.size(), f);
}

void dump_frame3(std::FILE *f, const nb_body3 *b, int n, int step)
{
    const char magic[8] = {'N', 'B', 'D', 'U', 'M', 'P', '3', 0};
    const uint64_t hdr[2] = {(uint64_t)n, (uint64_t)step};
    std::fwrite(magic, 1, 8, f);
    std::fwrite(hdr, sizeof(uint64_t), 2, f);
    std::vector<double> rec(7 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        double *r = &rec[7 * (size_t)i];
        r[0] = b[i].position.x;
        r[1] = b[i].position.y;
        r[2] = b[i].position.z;
        r[3] = b[i].velocity.x;
        r[4] = b[i].velocity.y;
        r[5] = b[i].velocity.z;
        r[6] = b[i].mass;
    }
    std::fwrite(rec.data(), sizeof(double), rec.size(), f);
}

// One frame as the reference draws it (storeFrame / scaleToWindow, src/gpu/BH/main.cu:28-60):
// pixel = position * W / NBODY_WIDTH + W / 2 (y not flipped), bodies of mass >= 1.6e29 (HBL) as
// red discs of radius 5, the others as white discs of radius 1, on black; later bodies overdraw
// earlier ones.  Written as a binary PPM instead of a video frame.
void write_ppm(const std::string &path, const nb_body *b, int n, int window)
{
    const double scale = (double)window / 10.0e11;  // WINDOW / NBODY_WIDTH (constants.h:21-22,31-32)
    std::vector<unsigned char> img((size_t)window * window * 3, 0);
    for (int i = 0; i < n; ++i) {
        const bool star = b[i].mass >= 1.6e29;
        const int r = star ? 5 : 1;
        const int cx = (int)(b[i].position.x * scale + window / 2), cy = (int)(b[i].position.y * scale + window / 2);
        for (int dy = -r; dy <= r; ++dy)
            for (int dx = -r; dx <= r; ++dx) {
                const int x = cx + dx, y = cy + dy;
                if (dx * dx + dy * dy > r * r || x < 0 || y < 0 || x >= window || y >= window)
                    continue;
                unsigned char *px = &img[((size_t)y * window + x) * 3];
                px[0] = 255;
                px[1] = star ? 0 : 255;
                px[2] = star ? 0 : 255;
            }
    }
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) {
        std::cerr << "cannot open " << path << std::endl;
        std::exit(EXIT_FAILURE);
    }
    std::fprintf(f, "P6\n%d %d\n255\n", window, window);
    std::fwrite(img.data(), 1, img.size(), f);
    std::fclose(f);
}

// checkArgs, src/gpu/BH/main.cu:62-84
bool checkArgs(int nBodies, int sim, int iter, int maxSim)
{
    if (nBodies < 1) {
        std::cout << "ERROR: need to have at least 1 body" << std::endl;
        return false;
    }
    if (sim < 0 || sim > maxSim) {
        std::cout << "ERROR: simulation doesn't exist" << std::endl;
        return false;
    }
    if (iter < 1) {
        std::cout << "ERROR: need to have at least 1 iteration" << std::endl;
        return false;
    }
    return true;
}

}  // namespace

int main(int argc, char **argv)
{
    std::string prog = argv[0];
    const size_t slash = prog.find_last_of('/');
    if (slash != std::string::npos)
        prog = prog.substr(slash + 1);
    enum { BH, DS, CPU } mode = prog == "DirectSum" ? DS : prog == "nBodyCPU" ? CPU : BH;

    int alg = mode == DS ? NB_ALG_DIRECT : NB_ALG_BARNES_HUT;
    unsigned seed = 12345;
    double theta = -1;
    std::string dump, ppm;
    int dump_every = 1, iters_flag = 300, window = 2048;
    bool quiet = false, energy = false;
    double thickness = -1;  // >= 0: three-dimensional run
    int integrator = NB_INT_EULER;
    int gpus = 1;
    bool exchange_cells = true;
    std::vector<const char *> pos;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                std::cerr << name << " needs a value" << std::endl;
                std::exit(-1);
            }
            return argv[++i];
        };
        if (a == "--alg") {
            std::string v = next("--alg");
            alg = v == "direct" ? NB_ALG_DIRECT : NB_ALG_BARNES_HUT;
        } else if (a == "--seed")
            seed = (unsigned)std::strtoul(next("--seed"), nullptr, 10);
        else if (a == "--theta")
            theta = std::atof(next("--theta"));
        else if (a == "--dump")
            dump = next("--dump");
        else if (a == "--ppm")
            ppm = next("--ppm");
        else if (a == "--window")
            window = std::atoi(next("--window"));
        else if (a == "--dump-every")
            dump_every = std::atoi(next("--dump-every"));
        else if (a == "--iters")
            iters_flag = std::atoi(next("--iters"));
        else if (a == "--integrator") {
            std::string v = next("--integrator");
            integrator = v == "leapfrog" ? NB_INT_LEAPFROG : NB_INT_EULER;
        } else if (a == "--energy")
            energy = true;
        else if (a == "--thickness")
            thickness = std::atof(next("--thickness"));
        else if (a == "--gpus")
            gpus = std::atoi(next("--gpus"));
        else if (a == "--exchange")
            exchange_cells = std::string(next("--exchange")) != "positions";
        else if (a == "--quiet")
            quiet = true;
        else
            pos.push_back(argv[i]);
    }

    int nBodies = mode == CPU ? 100 : 300;  // NUM_BODIES: cpu/constants.h:20, gpu/BH/constants.h:20
    int sim = 0, iters = iters_flag;
    if (pos.size() == 3) {
        nBodies = std::atoi(pos[0]);
        if (mode == CPU) {
            alg = std::atoi(pos[1]) == 0 ? NB_ALG_DIRECT : NB_ALG_BARNES_HUT;
            sim = std::atoi(pos[2]);
        } else {
            sim = std::atoi(pos[1]);
            iters = std::atoi(pos[2]);
        }
    }
    if (!checkArgs(nBodies, sim, iters, mode == BH ? 3 : 2))
        return -1;
    // the direct-sum and CPU programs number the solar system 2 (directSum_kernel.cu:274-286,
    // cpu/nBody.cpp:29-41); the library uses the Barnes-Hut program's numbering
    int scenario = sim;
    if (mode != BH && sim == 2)
        scenario = NB_SIM_SOLAR;
    if (scenario == NB_SIM_SOLAR)
        nBodies = 5;  // main.cu:101-102

    nb_params params;
    nb_default_params(&params);
    if (theta >= 0)
        params.theta = theta;
    params.integrator = integrator;
    if (dump_every < 1)
        dump_every = 1;

    // ---- one process per GPU ----
    // The ranks are started here, before anything in this process has touched the GPU.  Rank 0
    // makes the communicator id and hands it to the others through pipes.
    int rank = 0;
    unsigned char comm_id[NB_COMM_ID_BYTES];
    if (gpus < 1 || gpus > 64) {
        std::cout << "ERROR: --gpus must be 1..64" << std::endl;
        return -1;
    }
    if (gpus > 1) {
        if (!dump.empty() || energy) {
            std::cout << "ERROR: --dump and --energy need one process to hold every velocity; not with --gpus > 1" << std::endl;
            return -1;
        }
        std::vector<int> fds(2 * (size_t)gpus, -1);
        for (int r = 1; r < gpus; ++r)
            if (pipe(&fds[2 * (size_t)r]) != 0) {
                std::perror("pipe");
                return EXIT_FAILURE;
            }
        std::vector<pid_t> kids;
        bool child = false;
        for (int r = 0; r < gpus && !child; ++r) {
            const pid_t pid = fork();
            if (pid < 0) {
                std::perror("fork");
                return EXIT_FAILURE;
            }
            if (pid == 0) {
                child = true;
                rank = r;
            } else
                kids.push_back(pid);
        }
        // Every process keeps only the pipe ends it uses, so that a rank whose writer has gone
        // sees end-of-file instead of waiting for ever: the parent none, rank 0 the write ends,
        // rank r its own read end.
        for (int r = 1; r < gpus; ++r) {
            const bool keep_read = child && rank == r, keep_write = child && rank == 0;
            if (!keep_read)
                close(fds[2 * (size_t)r]);
            if (!keep_write)
                close(fds[2 * (size_t)r + 1]);
        }
        if (!child) {
            int failed = 0;
            for (pid_t pid : kids) {
                int st = 0;
                if (waitpid(pid, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0)
                    failed = 1;
            }
            return failed ? EXIT_FAILURE : 0;
        }
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < gpus) {
            if (rank == 0)
                std::cout << "ERROR: --gpus " << gpus << " but " << ndev << " GPU(s) are visible" << std::endl;
            return EXIT_FAILURE;
        }
        if (hipSetDevice(rank) != hipSuccess) {
            std::cerr << "hipSetDevice(" << rank << ") failed" << std::endl;
            return EXIT_FAILURE;
        }
        if (rank == 0) {
            // (a failure in here ends rank 0, which closes its write ends: the other ranks read
            // end-of-file and exit non-zero)
            nbody::check(nb_comm_unique_id(comm_id, sizeof(comm_id)), "nb_comm_unique_id");
            for (int r = 1; r < gpus; ++r) {
                if (write(fds[2 * (size_t)r + 1], comm_id, sizeof(comm_id)) != (ssize_t)sizeof(comm_id)) {
                    std::perror("write");
                    return EXIT_FAILURE;
                }
                close(fds[2 * (size_t)r + 1]);
            }
        } else {
            size_t got = 0;
            while (got < sizeof(comm_id)) {
                const ssize_t k = read(fds[2 * (size_t)rank], comm_id + got, sizeof(comm_id) - got);
                if (k <= 0)
                    break;  // end-of-file (rank 0 failed before it wrote) or an error
                got += (size_t)k;
            }
            close(fds[2 * (size_t)rank]);
            if (got != sizeof(comm_id)) {
                std::cerr << "rank " << rank << ": rank 0 ended before it sent the communicator id" << std::endl;
                return EXIT_FAILURE;
            }
        }
        quiet = quiet || rank != 0;
    }

    std::FILE *f = nullptr;
    if (!dump.empty()) {
        f = std::fopen(dump.c_str(), "wb");
        if (!f) {
            std::cerr << "cannot open " << dump << std::endl;
            return EXIT_FAILURE;
        }
    }
    if (thickness >= 0) {
        // the same loop over the three-dimensional object
        nbody::NBodySim3 nb3(nBodies, alg, &params, seed);
        nb3.setup(scenario, thickness);
        if (gpus > 1)
            nb3.commInit(rank, gpus, comm_id);
        std::vector<nb_body> flat;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i) {
            nb3.update();
            if ((f || !ppm.empty()) && ((i + 1) % dump_every == 0 || i + 1 == iters)) {
                nb3.readDeviceBodies();
                if (f)
                    dump_frame3(f, nb3.getBodies(), nBodies, i + 1);
                if (!ppm.empty() && rank == 0) {
                    flat.assign(nBodies, nb_body{});
                    for (int k = 0; k < nBodies; ++k) {
                        flat[k].mass = nb3.getBodies()[k].mass;
                        flat[k].position = {nb3.getBodies()[k].position.x, nb3.getBodies()[k].position.y};
                    }
                    char name[32];
                    std::snprintf(name, sizeof(name), "_%06d.ppm", i + 1);
                    write_ppm(ppm + name, flat.data(), nBodies, window < 16 ? 16 : window);
                }
            }
        }
        nb3.sync();
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (f)
            std::fclose(f);
        if (!quiet)
            std::cout << (alg == NB_ALG_DIRECT ? "direct sum" : "Barnes-Hut") << " 3-D N=" << nBodies << " sim=" << sim
                      << " iters=" << iters << " gpus=" << gpus << " : " << ms / iters << " ms/step" << std::endl;
        return 0;
    }
    nbody::NBodySim nb(nBodies, alg, &params, seed);
    nb.setup(scenario);
    if (gpus > 1) {
        nb.commInit(rank, gpus, comm_id);
        // (cell records while they are less data than the positions and the tree closes within the
        // levels they can be exchanged at; the library follows the bounding box and switches by itself)
        if (alg == NB_ALG_BARNES_HUT && exchange_cells && integrator == NB_INT_EULER &&
            nb_set_exchange(nb.handle(), NB_EXCHANGE_AUTO) != NB_OK && !quiet)
            std::cout << "note: " << nb_last_error() << "; exchanging positions" << std::endl;
    }
    auto print_energy = [&](const char *when) {
        double d[8];
        nb.diagnostics(d);
        std::cout << when << ": E=" << d[6] + d[7] << " K=" << d[6] << " U=" << d[7] << " P=(" << d[3] << ", " << d[4]
                  << ") L=" << d[5] << std::endl;
    };
    if (energy)
        print_energy("start");
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) {
        nb.update();
        if ((f || !ppm.empty()) && ((i + 1) % dump_every == 0 || i + 1 == iters)) {
            nb.readDeviceBodies();
            if (f)
                dump_frame(f, nb.getBodies(), nBodies, i + 1);
            // (every rank reads at the same steps: with the cell exchange the read gathers the
            // positions over the ranks; rank 0 draws)
            if (!ppm.empty() && rank == 0) {
                char name[32];
                std::snprintf(name, sizeof(name), "_%06d.ppm", i + 1);
                write_ppm(ppm + name, nb.getBodies(), nBodies, window < 16 ? 16 : window);
            }
        }
    }
    nb.sync();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (f)
        std::fclose(f);
    if (energy)
        print_energy("end");
    if (!quiet)
        std::cout << (alg == NB_ALG_DIRECT ? "direct sum" : "Barnes-Hut") << " N=" << nBodies << " sim=" << sim
                  << " iters=" << iters << " gpus=" << gpus << " : " << ms / iters << " ms/step" << std::endl;
    return 0;
}
