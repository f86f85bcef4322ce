// pack_rows.h — row numbers of the per-lane constant pack (float4[row][64]; see LossyDevTables in lossy_device.hpp).
// Shared by the host builder (tables.cpp, plain C++) and the device code. Rows below kPackRowsHot are what every frame
// of the 44.1 / 48 kHz chain kernels reads: lossy_chain2x_kernel copies only those into LDS; the rows behind them are
// read from global memory by the few tables that need them.
#pragma once

namespace flo {
constexpr int kRowWin = 0;        // 8 rows: window values of fold row r: (w[eo], w[oo], w[1024+eo], w[1024+oo])   mdct.rs:106-113
constexpr int kRowTw = 8;         // 4 rows: pre/post-rotation twiddles tw[lane + 64 r], two rows per float4           mdct.rs:81-86
constexpr int kRowF1 = 12;        // 4 rows: FFT pass-1 twiddles W512^(lane k), k = 1..7 (re, im pairs)
constexpr int kRowF2 = 16;        // 4 rows: FFT pass-2 twiddles W64^((lane & 7) k)
constexpr int kRowAth = 20;       // 4 rows: ATH amplitude thresholds of coefficients 16 lane .. 16 lane + 15
constexpr int kRowLane = 24;      // 1 row : (lane_bnd, lane_slot0, 1 / bins of band lane & 31, band slot range)
constexpr int kRowBo = 25;        // 4 rows: byte offset (8 x band) of each of the lane's 16 coefficients
constexpr int kRowKeep = 29;      // 4 rows: band-statistics restart multipliers (0.0 behind a band boundary, else 1.0)
constexpr int kRowDst = 33;       // 4 rows: byte offset of the slot each running (sum, max) is stored to (segment slot or trash)
constexpr int kRowLst = 37;       // 3 rows: byte offsets of the first 12 slots this band lane adds (zero slot when exhausted)
constexpr int kRowS10 = 40;       // 1 row : floats 0..23 = s10d[1..24] (spreading level per band distance), read uniformly
constexpr int kPackRowsHot = 41;
constexpr int kRowLstCold = 41;   // 3 rows: slots 12..23 of a band lane's list (tables with more than 24 slots per band)
constexpr int kPackRows = 44;
}  // namespace flo
