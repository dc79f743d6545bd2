// Host C++ classes with the reference's exact public signatures, implemented on top of the C ABI
// in mcgpu.h.  A build of bcosm/MonteCarloOptionsPricer that puts this include directory first and
// links libmcgpu.so instead of compiling src/models/RoughVolatility.cpp + src/models/LSMPricer.cpp
// gets the MI355X path engine with no source change in its driver
// (src/core/PredictionGen.cpp:566-570, :736-737, :790).  See INTEGRATION.md.
//
//   class RoughVolatility  <->  reference include/models/RoughVolatility.h:9-19
//   class LSM              <->  reference include/models/LSMPricer.h:5-15
//   PayoffFunction         <->  reference include/core/common.h:8-14
//   class AsymptoticAnalysis <-> reference include/models/AsymptoticAnalysisPricer.h:5-16
//   class MartingaleOptimization <-> reference include/models/MartingaleOptimizationPricer.h:7-18
//   class BranchingProcesses <-> reference include/models/BranchingProcessPricer.h:5-16
//
// Error behaviour mirrors the reference: std::runtime_error("Historical prices vector too small.")
// (RoughVolatility.cpp:317-319) and std::runtime_error("LSM::PredictOptionPrice: Empty pricePaths.")
// (LSMPricer.cpp:28-30).  Any device/runtime failure is also surfaced as std::runtime_error, which
// the reference driver already catches per row (PredictionGen.cpp:792-805).
// Objects are stateless and cheap to construct per row per thread, as the driver does; device
// state lives in a lazily created per-thread mcg_ctx.
#ifndef MCGPU_DROPIN_HPP
#define MCGPU_DROPIN_HPP

#include <algorithm>
#include <vector>

#ifndef MCGPU_PAYOFF_DEFINED
#define MCGPU_PAYOFF_DEFINED
inline double PayoffFunction(bool isCall, double stockPrice, double strike) {
    const double intrinsic = isCall ? stockPrice - strike : strike - stockPrice;
    return std::max(0.0, intrinsic);
}
#endif

class RoughVolatility {
public:
    RoughVolatility();
    // Estimates (xi, H, eta, rho) and S0 from the history on the host, then simulates
    // `path_num` rBergomi paths of `forward_steps` steps (dt = 1/252, r = 0.04) on the GPU.
    // Returns [path_num][forward_steps + 1], column 0 = last historical price.
    std::vector<std::vector<double>> GenerateStockPricePaths(const std::vector<double>& historical_prices,
                                                             int forward_steps, int path_num);
};

class LSM {
public:
    // Longstaff-Schwartz (value-iteration variant of the reference) on host-provided paths.
    double PredictOptionPrice(const std::vector<std::vector<double>>& pricePaths, double r, double strike,
                              double maturity, double dt, bool isCall, int polyOrder);
};

class AsymptoticAnalysis {
public:
    // Short-time asymptotic exercise boundary pricer (reference include/models/AsymptoticAnalysisPricer.h:5-16).
    // Returns 0.0 for empty or ragged input like the reference; throws
    // std::runtime_error("AsymptoticAnalysis: Volatility must be positive.") when sigma <= 0.
    double PredictOptionPrice(const std::vector<std::vector<double>>& pricePaths, double r, double strike,
                              double maturity, double dt, bool isCall, double sigma, double dividend);
};

class MartingaleOptimization {
public:
    // Primal/dual martingale bounds (reference include/models/MartingaleOptimizationPricer.h:7-18).  Throws
    // std::runtime_error("MartingaleOptimization: Empty pricePaths.") / ("... maxIterations must be positive.").
    double PredictOptionPrice(const std::vector<std::vector<double>>& pricePaths, double r, double strike,
                              double maturity, double dt, bool isCall, int polyOrder, int maxIterations = 5);
};

class BranchingProcesses {
public:
    // Lower/upper bound midpoint by branch resampling (reference include/models/BranchingProcessPricer.h:5-16).
    // Throws std::runtime_error("BranchingProcesses: Empty pricePaths." / "... No exercise times." /
    // "... Strike must be positive.") like the reference.
    double PredictOptionPrice(const std::vector<std::vector<double>>& pricePaths, double r, double strike,
                              double maturity, double dt, bool isCall, int numBranches,
                              const std::vector<int>& exerciseTimes);
};

#endif  // MCGPU_DROPIN_HPP
