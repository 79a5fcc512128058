// poisson_driver.cpp — the replacement for Poissons_SYCL.cpp's main() (PS:658-731):
// same sequence (build the level hierarchy, build the load vector, run
// fullmultigrid on the finest level, print the solution size), same function
// names, on libmgx.  Adds what SURVEY D10 notes the reference never prints:
// the residual norms and timing.
//
//   poisson_driver [finest=10] [coarsest=7] [mu0=30] [mu1=10] [mu2=10] [f32|f64] [n_gpus=1]
//
// Defaults are the reference's compile-time globals (PS:17-22, 123, 127).  n_gpus > 1 splits the
// finest levels into row slabs, one per GPU (mgx_config.n_gpus; MGX_DRIVER_DEVICES="0,0,1,1" places
// the slabs explicitly - several may share a device); the program is the same: fullmultigrid (PS:727).
#include "mgx_reference_api.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

using namespace mgxref;

template <typename Real> static int run(const parameters& prm)
{
    queue q;                                                                  // PS:659
    // creating the level data for the different levels (PS:661-690)
    std::vector<matrix_elements_for_jacobi>& jacobi_matrices = build_hierarchy<Real>(prm);

    std::vector<Real> f_global = globalforcefunction<Real>();                 // PS:725
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<Real> solution_finest =
        fullmultigrid(q, jacobi_matrices[jacobi_matrices.size() - 1], f_global);   // PS:727
    const auto t1 = std::chrono::steady_clock::now();
    std::cout << "Size of finest level solution is " << solution_finest.size() << "\n";   // PS:728

    // D10: report the residual of what fullmultigrid returned, then iterate
    // V-cycles to 1e-8 the way MF:193 multigrid_solver would be used.
    double r_fmg = 0.0;
    check(mgx_residual_norm(jacobi_matrices.back().handle, prm.finest_level, &r_fmg), jacobi_matrices.back().handle,
          "mgx_residual_norm");
    const std::size_t n = std::size_t(mgx_level_n(prm.finest_level));
    std::printf("fullmultigrid: %.3f ms, ||b - A u||_2 = %.6e, u(1/2,1/2) = %.7f\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), r_fmg,
                double(solution_finest[(n / 2) * n + n / 2]));
    mgx_stats st{};
    std::vector<double> hist;
    std::vector<Real> u = multigrid_solver(f_global, 1e-8, 60, &st, &hist);
    std::printf("solve to 1e-8: %d cycles (FMG pass + V-cycles) on %d GPU%s, %.3f ms, ||r||/||r0|| = %.3e, %.3e fine-grid updates/s\n",
                st.cycles, prm.n_gpus, prm.n_gpus > 1 ? "s (row slabs)" : "", st.seconds * 1e3, st.final_residual / st.initial_residual,
                st.seconds > 0 ? st.fine_updates / st.seconds : 0.0);
    for (std::size_t k = 0; k < hist.size(); ++k) std::printf("  cycle %2zu  ||r||_2 = %.6e\n", k, hist[k]);
    std::cout << "Program Running Correctly ";                                // PS:729
    std::cout << std::endl;
    return 0;
}

int main(int argc, char** argv)
{
    parameters prm;
    if (argc > 1) prm.finest_level = std::atoi(argv[1]);
    if (argc > 2) prm.coarsest_level = std::atoi(argv[2]);
    if (argc > 3) prm.mu0 = std::atoi(argv[3]);
    if (argc > 4) prm.mu1 = std::atoi(argv[4]);
    if (argc > 5) prm.mu2 = std::atoi(argv[5]);
    const bool f32 = (argc > 6 && std::strcmp(argv[6], "f32") == 0);
    if (argc > 7) prm.n_gpus = std::atoi(argv[7]);
    if (const char* dv = std::getenv("MGX_DRIVER_DEVICES")) {
        int i = 0;
        for (const char* p = dv; *p && i < MGX_MAX_GPUS; ++i) {
            prm.devices[i] = int(std::strtol(p, const_cast<char**>(&p), 10));
            if (*p == ',') ++p;
        }
    }
    try {
        return f32 ? run<float>(prm) : run<double>(prm);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "poisson_driver: %s\n", e.what());
        return 1;
    }
}
