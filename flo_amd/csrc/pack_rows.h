// pack_rows.h — row numbers of the per-lane constant pack (float4[row][64]; see LossyDevTables in lossy_device.hpp).
// Shared by the host builder (tables.cpp, plain C++) and the device code. The rows are ordered by who reads them:
//   rows below kPackRowsHotT : what the transform wave of lossy_chain2q_kernel reads every frame (its LDS copy);
//   rows below kPackRowsHot  : plus the quantiser rows of the forms whose transform wave also quantises;
//   the rows behind them are read from global memory by the few tables / kernels that need them.
#pragma once

namespace flo {
constexpr int kRowWin = 0;        // 8 rows: window values of fold row r: (w[eo], w[oo], w[1024+eo], w[1024+oo])   mdct.rs:106-113
constexpr int kRowTw = 8;         // 4 rows: pre/post-rotation twiddles tw[lane + 64 r], two rows per float4           mdct.rs:81-86
constexpr int kRowF1 = 12;        // 4 rows: FFT pass-1 twiddles W512^(lane k), k = 1..7 (re, im pairs)
constexpr int kRowF2 = 16;        // 4 rows: FFT pass-2 twiddles W64^((lane & 7) k)
constexpr int kRowLane = 20;      // 1 row : (lane_bnd, lane_slot0, 1 / bins of band lane & 31, band slot range)
constexpr int kRowKeep = 21;      // 4 rows: band-statistics restart multipliers (0.0 behind a band boundary, else 1.0)
constexpr int kRowDst = 25;       // 4 rows: byte offset of the slot each running (sum, max) is stored to (segment slot or trash)
constexpr int kRowLst = 29;       // 3 rows: byte offsets of the first 12 slots this band lane adds (zero slot when exhausted)
constexpr int kRowS10 = 32;       // 1 row : floats 0..23 = s10d[1..24] (spreading level per band distance), read uniformly
constexpr int kPackRowsHotT = 33;
constexpr int kRowAth = 33;       // 4 rows: ATH amplitude thresholds of coefficients 16 lane .. 16 lane + 15
constexpr int kRowBo = 37;        // 4 rows: byte offset (8 x band) of each of the lane's 16 coefficients
constexpr int kPackRowsHot = 41;
constexpr int kRowLstCold = 41;   // 3 rows: slots 12..23 of a band lane's list (tables with more than 24 slots per band)
// natural layout of the packer wave (lane l, block k = positions 128 k + 2 l and + 1): entry 2 k + j of the lane's sixteen
constexpr int kRowAthN = 44;      // 4 rows: ATH amplitude thresholds of those positions
constexpr int kRowBoN = 48;       // 4 rows: byte offset (16 x band) of those positions into a float4-per-band table
constexpr int kRowBlk = 52;       // 1 row : dwords 0..7 = bit b set when block k (positions 128 k .. 128 k + 127) holds bins of band b; read uniformly
constexpr int kPackRows = 53;
}  // namespace flo
