// cpk_cells.inl -- the reference's cell-level primitives (inc/pairwiseAligner.h:186-237: cell_calculateForward /
// cell_calculateBackward, diagonalCalculationForward / Backward, the straddle step of diagonalCalculationTotalProbability,
// the posterior of diagonalCalculationPosteriorMatchProbs) for the drop-in layer's DpDiagonal / DpMatrix containers.
// Part of the single HIP translation unit cpecan_kernels.hip; not compiled on its own.
//
// These are what the reference's UNIT TESTS link (tests/pairwiseAlignerTest.c:155-324); nothing on the hot path calls them.
// The DP arithmetic still runs on the GPU (the library holds no CPU implementation of it): the host hands over a flat
// buffer of cells and a list of operations, ONE lane applies them in the reference's order -- the forward form gathers
// into `current`, the backward form scatters from `current` into its three neighbours (impl/pairwiseAligner.c:382-395),
// cells in ascending x-y (:609-624), transitions in list order (impl/stateMachine.c:450-480, :689-714) -- with the
// kernels' own logAdd.
constexpr int kCellsForward = 0, kCellsBackward = 1, kCellsPosterior = 2;

__device__ __forceinline__ void cell_transition(const Cubic *lg, int backward, double *from, double *to, int f, int t, double w) {
    if (backward) from[f] = logadd(lg, from[f], to[t] + w);  // doTransitionBackward, impl/pairwiseAligner.c:392-395
    else to[t] = logadd(lg, to[t], from[f] + w);             // doTransitionForward, :382-385
}

__global__ void __launch_bounds__(CPK_WAVE) cpecan_ref_cells(const CpkModel m, int mode, const CpkCellOp *ops, int n, double *buf, double total) {
    __shared__ __attribute__((aligned(16))) double lds[kLdsCubics];
    fill_cubics(lds);
    __syncthreads();
    if (threadIdx.x != 0) return;
    const Cubic *lg = reinterpret_cast<const Cubic *>(lds);
    const int S = m.nStates;
    for (int i = 0; i < n; i++) {
        const CpkCellOp op = ops[i];
        if (mode == kCellsPosterior) {
            // exp((F.match + B.match) - total), impl/pairwiseAligner.c:683-685; cur = the forward cell, lower = the backward cell
            buf[op.upper] = exp((buf[op.cur] + buf[op.lower]) - total);
            continue;
        }
        double *cur = buf + op.cur;
        double *lower = op.lower >= 0 ? buf + op.lower : nullptr, *middle = op.middle >= 0 ? buf + op.middle : nullptr,
               *upper = op.upper >= 0 ? buf + op.upper : nullptr;
        const double eM = m.matchEm[op.cX * 5 + op.cY], eX = m.gapXEm[op.cX], eY = m.gapYEm[op.cY];
        const int b = mode == kCellsBackward;
        if (S == 5) {  // impl/stateMachine.c:450-480
            if (lower) {
                cell_transition(lg, b, lower, cur, 0, 1, eX + m.shortOpenX);
                cell_transition(lg, b, lower, cur, 1, 1, eX + m.shortExtendX);
                cell_transition(lg, b, lower, cur, 0, 3, eX + m.longOpenX);
                cell_transition(lg, b, lower, cur, 3, 3, eX + m.longExtendX);
            }
            if (middle) {
                cell_transition(lg, b, middle, cur, 0, 0, eM + m.matchContinue);
                cell_transition(lg, b, middle, cur, 1, 0, eM + m.matchFromShortX);
                cell_transition(lg, b, middle, cur, 2, 0, eM + m.matchFromShortY);
                cell_transition(lg, b, middle, cur, 3, 0, eM + m.matchFromLongX);
                cell_transition(lg, b, middle, cur, 4, 0, eM + m.matchFromLongY);
            }
            if (upper) {
                cell_transition(lg, b, upper, cur, 0, 2, eY + m.shortOpenY);
                cell_transition(lg, b, upper, cur, 2, 2, eY + m.shortExtendY);
                cell_transition(lg, b, upper, cur, 0, 4, eY + m.longOpenY);
                cell_transition(lg, b, upper, cur, 4, 4, eY + m.longExtendY);
            }
        } else {  // :689-714
            if (lower) {
                cell_transition(lg, b, lower, cur, 0, 1, eX + m.shortOpenX);
                cell_transition(lg, b, lower, cur, 1, 1, eX + m.shortExtendX);
                cell_transition(lg, b, lower, cur, 2, 1, eX + m.shortSwitchToX);
            }
            if (middle) {
                cell_transition(lg, b, middle, cur, 0, 0, eM + m.matchContinue);
                cell_transition(lg, b, middle, cur, 1, 0, eM + m.matchFromShortX);
                cell_transition(lg, b, middle, cur, 2, 0, eM + m.matchFromShortY);
            }
            if (upper) {
                cell_transition(lg, b, upper, cur, 0, 2, eY + m.shortOpenY);
                cell_transition(lg, b, upper, cur, 2, 2, eY + m.shortExtendY);
                cell_transition(lg, b, upper, cur, 1, 2, eY + m.shortSwitchToY);
            }
        }
    }
}
