// dp.h -- device data model and scalar helper functions of the GHMM decode kernels.
//
// Everything here is written once and compiled twice: by hipcc for gfx950 (the product), and by g++ with
// -DAUGX_EMU for the lane-loop emulator under tests/emu (a debugging aid for a GPU-less container; it is
// test infrastructure, never a product fallback).  All arithmetic on the decode path is fp64 add/compare
// plus uint64 fixed-point sums, compiled with -ffp-contract=off, so both builds are bit-identical.
//
// The per-state formulas restate the reference scorers (cited per function); data layout, candidate lists
// and the kernel decomposition are ours.
#pragma once
#include <stdint.h>
#include <math.h>
#include "../../../include/augx.h"

#ifdef AUGX_EMU
#define AUGX_HD inline
#else
#include <hip/hip_runtime.h>
#define AUGX_HD __host__ __device__ __forceinline__
#endif

// a table pointer read from a struct is generic to the compiler, and a load through it a flat load (slower, and it holds up the
// LDS wait counter as well): the tables live in HBM, say so
#if defined(__HIP_DEVICE_COMPILE__)
#define AUGX_GTAB(p) (p) /* (the table pointers carry their address space themselves: TabPtr below) */
#else
#define AUGX_GTAB(p) (p)
#endif

namespace augx {
namespace dev {

constexpr int WAVE = 64;
constexpr int SP = 48;          // column stride of the trellis storage (S <= SP)
constexpr int SPX = AUGX_MAX_STATES; // column stride bound of the dense kernels (dense.h: models with UTR states, S = 71)
constexpr int NUFX = 6;         // UTR content prefix fields: 5' single/initial fwd, rev; 5' internal/terminal fwd, rev; 3' fwd, rev
constexpr int UFX_5IF = 0, UFX_5IR = 1, UFX_5F = 2, UFX_5R = 3, UFX_3F = 4, UFX_3R = 5;
constexpr int NUCNT = 4;        // UTR site counts: forward TSS windows, forward stop codons, reverse poly-A boxes, reverse start codons
constexpr int UCNT_TF = 0, UCNT_FS = 1, UCNT_TM = 2, UCNT_RT = 3;
constexpr int NUSIG = 4;        // per-base UTR signal record: tssProb fwd (window begins here), tssProb rev (window ends here), ttsProbPlus / Minus (box begins here)
constexpr int USIG_TSSF = 0, USIG_TSSR = 1, USIG_TTSP = 2, USIG_TTSM = 3;
constexpr int NFX = 20;         // fixed-point prefix fields per slot: [strand 2][phase 3][table 3], inF, inR
constexpr int FX_INF = 18, FX_INR = 19;
constexpr int NSIG = 10;        // per-position signal record: eIg eIn dssF dssR assF assR tisF tisR eqD stopF
constexpr int SIG_EIG = 0, SIG_EIN = 1, SIG_DSSF = 2, SIG_DSSR = 3, SIG_ASSF = 4, SIG_ASSR = 5, SIG_TISF = 6, SIG_EUIN = 6 /* (SIG_TISF is not in use) */, SIG_TISR = 7, SIG_EQD = 8, SIG_STOPF = 9;
constexpr int CHUNK = 1024;     // slots per scan chunk; every piece is padded to a multiple of CHUNK
constexpr int LONG_RING = 1024; // ring depth for the states consumed at lag dStateLen (must exceed it)
constexpr uint16_t BP_NONE = 0xFFFF;
// LDS window sizes (powers of 2).  The trellis' windows (igenic column, list values, staged candidates) may be set at build time
// (experiments with the LDS footprint of kTrellis: what does not fit a window is read back from HBM, the paths exist and are tested)
#ifndef AUGX_VIG_WIN
#define AUGX_VIG_WIN 512
#endif
#ifndef AUGX_LIST_WIN
#define AUGX_LIST_WIN 256
#endif
#ifndef AUGX_ITEM_CAP
#define AUGX_ITEM_CAP 2048
#endif
constexpr int CODE_WIN = 1024, NS_WIN = 128, CNT_WIN = 1024, VIG_WIN = AUGX_VIG_WIN, LIST_WIN = AUGX_LIST_WIN, FX_WIN = 128, ATG_WIN = 64;
constexpr int LEN_WIN = 384;    // exon length-distribution entries cached in LDS (longer exons read the HBM table)
constexpr int LENI_MAX = 1024;  // intron length distribution cached in LDS when d < LENI_MAX

#define AUGX_NINF (-INFINITY)
#define AUGX_EXACT_LIMIT 4194304.0 /* 2^(53 - AUGX_Q_BITS): below it every sum of model terms is exact in fp64 */

// one candidate of a variable-length state: everything but the predecessor's Viterbi value (kCandidates -> kTrellis)
struct Item {
    double te;      // ln(transition * emission) of the candidate, -inf if infeasible
    uint32_t kp;    // [31:22] the (base, state) pair: (base offset in the block << 6) | state, [21:0] tie-break key = eop + KEY_BIAS
    uint32_t src;   // where the predecessor value lives: [31:30] tag, [29:28] ancestor index, [27:0] payload
};
// two candidates of a cell closer than this in ln are a NEAR TIE: the model terms are rounded to 2^-31 once, a few hundred
// of them can move the difference of two alternative paths by up to ~1e-7, so the reference -- which rounds differently -- may decide such a
// cell the other way.  The back-trace counts the cells on the chosen path where that can have happened.
constexpr double AUGX_NEAR_TIE = 2e-7;
constexpr int KEY_BITS = 22;                 // pieces on the device path are shorter than 2^22 bases
constexpr uint32_t KEY_MASK = (1u << KEY_BITS) - 1;
constexpr int KEY_BIAS = 64;                 // key = eop + KEY_BIAS >= 0 (eop >= -(3 + W) - 1 for a left-truncated initial exon)
constexpr uint32_t SRC_LIST = 0, SRC_VIG = 1, SRC_COL0 = 2; // tags; LIST payload: [27:26] list, [25:24] frame, [23:0] entry
// bases per trellis block (template parameter BLK of the candidate and trellis kernels): smaller than every lag of the
// model except the lag-1 chain states.  8 where the species' windows allow it, else 4 or 2 (layout.h: chooseBlockSize)
constexpr int MAXPL = AUGX_MAX_CLASSES;      // GC-content classes inside one piece: as many as a model may have
#ifndef AUGX_MAXPL_LDS
#define AUGX_MAXPL_LDS 8                      /* (tests build the emulator with 1 to exercise the other path) */
#endif
constexpr int MAXPL_LDS = AUGX_MAXPL_LDS;    // planes whose transition terms the candidate kernel keeps in LDS (the others: L2)
constexpr int MAXNB = 32;                    // blocks per tile of 64 bases at the smallest block size (2)

struct CandAlloc;
// ---- segment-parallel trellis ("segments").  A piece is cut into segments of whole tiles.  Pass 1 runs every
// segment at once: segment 0 from the true start, segment k >= 1 "dead" -- nothing alive before its first base but the synch
// state at value 0 -- so that its values are those of the true run plus an unknown constant once the two have converged
// (every addition on the path is exact, include/augx.h AUGX_Q_BITS: a common constant commutes with the whole recurrence).
// Pass 2 re-runs the head of every segment k >= 1 from the end state of segment k-1 and compares what it retires (igenic
// column, long-lag cells, list values, the full column at every tile end) with what pass 1 left: when `segCheckTiles`
// consecutive tiles differ by one constant D, everything after is the old value + D and the pass stops.  A fix-up that
// does not get there before `tlim` gives up; pass 3 then continues that piece sequentially from the give-up point.
struct SegDesc {
    int32_t piece, k;      // piece, index of the segment inside the piece
    int32_t t0, t1;        // tiles [t0, t1) of the piece
    int32_t tlim;          // last tile the fix-up of this segment may rewrite (so that it stays clear of what fix-up k+1 reads)
    int32_t pad;
};
// one possible start of a short intron (entry of the LD / RD candidate lists): everything a lessD candidate needs of it,
// in one 16-byte record (position, the two bases before the splice site, intron-content prefix at the position)
struct IntronStart { int32_t pos; uint32_t ctx; uint64_t fx; };
// one possible begin of a UTR exon (entry of the TF / LA / FS / LR / TM / RT site lists, dense.h): the end of the predecessor state
// and, per content model that can follow, (ln begin signal) - (content prefix before the first base of the middle part)
struct USite { int32_t pos; int32_t pad; double b[3]; };
// an acceptor site whose value changes while the sweep goes on (the reference's aSSProb memo is emptied and the site computed again
// under another GC class, assmemo.h): laSite[i].pad - 1 is the first of its entries here, ascending by key; a candidate of
// (end base j, state s) adds `cum` of the last entry with key <= (j << 7 | s) to the site's b value
struct LaSw { uint32_t key; uint32_t more; double cum; };
// one acceptor site whose value the replay of the memo found to come from another class than the one k1SiteSignals took (entry li of
// the LA list): the longass states of its column get the value of plane longPl (-1: as it is), the UTR exons that begin at it the
// value of plane basePl and, from the calls sw[swOff .. swOff + nSw) on, those of the planes named there (dense.h: k1AssPatch)
struct AssPatch { int32_t li; int8_t longPl, basePl; int16_t nSw; int32_t swOff; };
struct AssSwIn { uint32_t key; int32_t pl; };
// flat model tables on the device (pointers are device pointers; in the emulator, host pointers)
// Pointers to the MODEL TABLES (ln probabilities of the species' parameter files, read-only for the lifetime of a decoder).  In
// device code they point into the constant address space: a table look-up may then be moved across the kernel's own stores and
// several look-ups be in flight together -- with a generic pointer the compiler must assume that a store to a record array
// changes the tables (kSignals waited 78 % of its time for single loads in flight: 14.4 -> 9.2 ms when its stores moved to the
// end of the function, round 5).  Same size and layout on the host.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(AUGX_EMU)
typedef const __attribute__((address_space(4))) double *TabPtr;
#else
typedef const double *TabPtr;
#endif
struct DevTables {
    int kIn, NPin;             // order and table length of the intron content table (augx_tables::k_in; = k, NP but for one species)
    int S, C, k, NP, W, U, As, Ae, Ds, De, Li, Le, d, dStateLen, max_exon_len, min_exon_len;
    int tis_n, tis_k, ass_n, ass_k, tis_nbins, tis_mem, synch, gc_win, gc_weighing_type;
    int dssGc;                 // a donor site may read gc as well as gt (/IntronModel/allow_dss_consensus_gc); dss_pat has a second half for them
    int soft;                  // soft-masking: lower-case bases carry lnSoft on igenic and intron states
    double lnSoft;
    int kind[AUGX_MAX_STATES], win[AUGX_MAX_STATES], type[AUGX_MAX_STATES], reachable[AUGX_MAX_STATES];
    int n_anc[AUGX_MAX_STATES], anc[AUGX_MAX_STATES][AUGX_MAX_ANC];
    double ln_init[AUGX_MAX_STATES], ln_term[AUGX_MAX_STATES];
    double ln_startcodon[64];
    double ln_stop_ochre, ln_stop_amber, ln_stop_opal, ln_quarter, ln_n_coding, ln4, ass_pat_invalid;
    double gc_zus[AUGX_MAX_CLASSES][4], gc_weight_matrix[16];
    // untranslated regions (include/augx.h: the utr block of augx_tables)
    int utr, tss_upwin, tss_start, tss_end, tata_start, tata_end, d_tss_tata_min, d_tss_tata_max, dpc, boxlen, tts_spacing;
    unsigned long long startMask; // augx_tables::start_mask: the codons that may start a gene under the translation table
    int stopMask;              // augx_tables::stop_mask: bit 0 taa, bit 1 tag, bit 2 tga end a reading frame
    int uk, uNP;               // order and table length of the UTR exon content tables (augx_tables::utr_k)
    int uML, uM3S, uM3T, tssup_k, tss_n, tss_k, tsstata_n, tsstata_k, tata_n, tata_k, tts_n, tts_k;
    double ln_tts_rand, ln2;
    TabPtr utr5init_emi, utr5_emi, utr3_emi, tssup_emi, tss_motif, tsstata_motif, tata_motif, tts_motif, aataaa,
        len5s, len5i, len5n, len5t, len3s, len3i, len3n, len3t, tail5s, tail3s;
    int dense;                 // the model is decoded by the dense kernels (dense.h)
    int vbit[AUGX_MAX_STATES]; // bit of a variable-length state in the end-gate mask (the state index itself while S <= 64)
    int uvS[16], nUv;          // the exon-like UTR states, ascending (dense.h: slot of a state in the descriptor kernel)
    TabPtr ln_trans, ig_emi, ig_short, in_emi, ex_emi, ex_init, ex_et, ex_pls, tis_motif, ass_motif,
        tis_bin_bounds, tis_bin_ln, ass_pat, dss_pat, len_intron, len_single, len_initial, len_internal,
        len_terminal;
};

// a batch of pieces laid out in one slot space.  Piece p owns slots [off[p], off[p+1]); slot off[p] is the
// "before the first base" slot, base q lives in slot off[p]+1+q; off[p] is a multiple of CHUNK.
struct BatchView {
    int nPieces;
    int64_t N;                 // total slots
    int nChunks;
    const int64_t *off;        // [nPieces+1]
    const int32_t *len;        // [nPieces]
    const int32_t *initKind, *termKind;
    const int32_t *chunkPiece; // [nChunks]
    int32_t *cls;              // [nPieces] GC class of plane 0 of the piece (-1: only while the content stairs are being settled)
    int32_t *clsMinMax;        // [nPieces][2]
    // GC-content classes inside a piece (reference NAMGene::viterbiAndForward switches all class-dependent tables at every
    // step of the content stairs, src/namgene.cc:245-248: a state ENDING at base j is scored with the tables of class(j)).
    // The classes of a piece are numbered by first appearance ("planes"); every class-dependent prefix / list array exists
    // once per plane ([nPl][...], plane-major), and the end base of a state selects the plane.
    int nPl;                   // planes allocated in this batch (1: no multi-class piece)
    int64_t listCap;           // entries per plane of the candidate-side list arrays
    const int64_t *listOffs;   // [nPieces+1] first entry of each piece in the list arrays: the lists are sized from the counted
                               // sites (the longest of the piece's six lists, padded), not from the worst case of one per two bases
    int32_t *listCnt;          // [nPieces] entries of the longest list of the piece (K1 -> host)
    uint8_t *gcRaw;            // [N] class of the GC window STARTING at this base (unsmoothed; input of the stairs)
    uint8_t *gcPlane;          // [N] plane of the base
    int32_t *nPlanes;          // [nPieces]
    int32_t *planeCls;         // [nPieces][MAXPL] class of each plane
    const char *raw;           // [N] ASCII (slot layout)
    uint8_t *code;             // [N] 0..3 acgt, 4 invalid / padding
    uint32_t *cnt;             // [N][NCNT] prefix counts (32 bits: a piece is shorter than 2^22 bases)
    uint32_t *nsm;             // [N][6] prefix max of (stop position+1) per residue class, fwd 0..2, rev 3..5
    uint64_t *fx;              // [N][NFX] fixed-point prefix sums
    double *sig;               // [N][NSIG]
    uint64_t *gate;            // [N] bit s: variable-length state s passes its end gate at this base
    int32_t *site;             // [N][4] list index of the splice-site candidate ending here (LA, LR, LD, RD) or -1
    uint64_t *chunkTot;        // scratch [nChunks][NFX]
    // untranslated regions (dense.h; allocated for models with UTR states only)
    uint64_t *ufx;             // [N][NUFX] fixed-point prefix sums of the UTR content models; the term of a base comes from the class OF THAT BASE
    uint32_t *ucnt;            // [N][NUCNT] prefix counts of the UTR begin sites
    double *usig;              // [N][NUSIG]
    USite *tfSite, *laSite, *fsSite, *lrSite, *tmSite, *rtSite; // [listCap] begin-site lists (laSite / lrSite run parallel to laPos / lrPos)
    const LaSw *laSw;          // acceptor sites whose value changes during the sweep (NULL: none)
    const double *tss0;        // [nPieces][2] (NULL: none; NaN: none for that piece) the value of the TSS window that begins at base 0 of the piece,
                               // forward / reverse, where the reference answers it from what an EARLIER sequence of the same length left in
                               // tssProbsPlus[0] / tssProbsMinus[0] (dense.h: k1UtrSignals; include/augx.h: augx_tss0)
    uint8_t *bpD;              // [N][S] back pointers of the chain and fixed-lag states (ancestor index, 0xFF: none)
    struct UDesc *ud;          // [udCap] descriptors of the open (end base, UTR exon state) pairs, the pairs of a block contiguous (kUtrDesc)
    int64_t udCap;
    uint64_t *udOff;           // [nBlk] first descriptor of the block
    uint32_t *udCnt;           // [nBlk] descriptors of the block
    // trellis
    uint16_t *bp;              // [N][SP] back pointers
    uint8_t *bpChain;          // [N][8] the back pointers of the (at most 8) single-base chain states once more, one byte each, in
                               // the order of their state index: the back-trace walks their long runs 256 bases per step through it
    double *cells;             // [N][S] dense ln V (debug/test only) or NULL
    double *fwd;               // [N][S] dense ln of the forward variables (only when posterior sampling is asked for) or NULL
    double *lnFwd;             // [nPieces] ln P(sequence) = the sum over all paths
    int32_t *nearTie;          // [nPieces] arg-max decisions on the chosen path whose runner-up lies within AUGX_NEAR_TIE of the winner (AUGX_TIMING / AUGX_NEAR_TIES) or NULL
    uint64_t *prof;            // [nPieces][4][8] cycle counters of the trellis wavefronts (AUGX_PROF=1) or NULL
    double *vig;               // [N] ln V of the igenic state (gathered by start-codon / reverse-stop candidates)
    double *longV;             // [N][6] ln V of longdss_f (0..2) and rlongass_f (3..5): read back at lag dStateLen by equalD
    int32_t *laPos; double *laVal;   // forward acceptor candidates  (longass_f live):  [N/2] , [N/2][3]
    int32_t *lrPos; double *lrVal;   // reverse donor candidates     (rlongdss_f live)
    struct IntronStart *ldEnt; double *ldVal;   // forward short-intron starts  (longdss_f live)
    struct IntronStart *rdEnt; double *rdVal;   // reverse short-intron starts  (rlongass_f live)
    int32_t *atgPos;                 // start codons (position of the a of atg) [N/2]
    // candidate-side constants of the list entries (independent of the Viterbi values: written by the prep kernels)
    double *laPls; uint64_t *laFx;   // [cap][3] per phase a: ln P_ls of the first k bases; exon-content prefix at bs+k-1
    double *lrEt; uint64_t *lrFx;    // [cap][3] per phase a: exon-terminal content of bs..bs+Le-1; exon-content prefix at bs+Le-1
    double *atgD; uint64_t *atgFx;   // [cap][3] (begin part, ln P_ls, initial content), [cap] exon-content prefix at bs+k-1+Li
    int32_t *rsPos; double *rsBegin; uint64_t *rsFx; // reverse stop codons: position, ln stop prob, [cap][3] exon-content prefix at bs-1
    double *plsR;                    // [N][3] reverse strand: ln P_ls of the k bases ending at this base, per frame
    // candidates of the variable-length states, grouped by block of BLK bases (block index = off[p]/BLK + j/BLK)
    int blk;                         // block size of this batch (8 or 4)
    int64_t nBlk;                    // N / blk
    uint32_t *blkCnt;                // [nBlk][2] (pairs, items) of the block
    uint32_t *blkSplit;              // [nBlk][3] items of the block up to the pair boundaries near 1/3 and 2/3 / of all states but RTERMINAL (they come first)
    uint64_t *blkOff;                // [nBlk][2] first pair / first item of the block (the blocks of a tile are contiguous)
    struct CandAlloc *candAlloc;     // running totals of pairs / items handed out to tiles
    Item *items;                     // [items]
    int64_t itemCap;
    // segments of the trellis (see SegDesc)
    int nSegs;                 // >= nPieces; == nPieces: no piece is cut
    const SegDesc *segs;       // [nSegs] grouped by piece, ascending k
    const int32_t *pieceSeg0;  // [nPieces+1] first segment of each piece
    int segCheckTiles;         // consecutive verified tiles that end a fix-up: they cover the longest look-back of the model
    int32_t *tileMinEop;       // [N/WAVE] smallest predecessor position (eop) of any live candidate that ends in the tile (K2a; INT_MAX: none)
    int32_t *tileCross;        // [N/WAVE] ... of any candidate that ends in one of the next segCheckTiles tiles: what the future can still read
    double *ckRing;            // [nSegs][2][WAVE][SP] the ring at the end of pass 1 of the segment [0] / where its fix-up gave up [1]
    double *ckCol;             // [N/WAVE][SP] ln V column at the end of every tile (pass 1; compared and replaced by pass 2)
    int32_t *segStop;          // [nSegs] pass 2: last tile rewritten when converged (>= t0 - 1), -2 - tile when it gave up after `tile`
    double *segD;              // [nSegs] pass 2: (value in the frame of segment k-1) - (value in the frame of segment k)
    int32_t *segStop2;         // [nSegs] pass 3: the continuation of a fix-up that gave up converged after this tile (-1: none; last tile of the piece: ran to the end)
    double *segD2;             // [nSegs] its offset
    int32_t *pieceCovered;     // [nPieces] last tile of the piece a continuation has redone (-1: none)
    int32_t *segStatus;        // [nSegs] abort flags
    int32_t *brkPos;           // [nSegs] kSegFinalize: region r of the piece = bases (brkPos[r-1], brkPos[r]] ...
    double *brkOff;            // [nSegs] ... whose stored values are true value - brkOff[r]
    // results
    double *lnv;               // [nPieces]
    int32_t *status;           // [nPieces]
    int32_t *finalState;       // [nPieces]
    int32_t *pathRec;          // [N/8 + 64*nPieces][3]  (begin, end, state), reverse order per piece
    int32_t *pathCount;        // [nPieces]
};

AUGX_HD int mod3(int k) { return k >= 0 ? k % 3 : (k % 3 + 3) % 3; }
AUGX_HD bool isUtrIntronKind(int k) { return k == AUGX_K_UTR5INTRON || k == AUGX_K_UTR3INTRON || k == AUGX_K_RUTR5INTRON || k == AUGX_K_RUTR3INTRON; }
AUGX_HD bool isUtrIntronVarKind(int k) { return k == AUGX_K_UTR5INTRONVAR || k == AUGX_K_UTR3INTRONVAR || k == AUGX_K_RUTR5INTRONVAR || k == AUGX_K_RUTR3INTRONVAR; }
AUGX_HD bool isUtrExonKind(int k) { return k >= AUGX_K_UTR5SINGLE && k <= AUGX_K_RUTR3TERM && !isUtrIntronKind(k) && !isUtrIntronVarKind(k); }
// scan-field arrays (cnt, nsm, fx) are stored chunk-major, field-major inside a chunk: [chunk][field][CHUNK],
// so that the prefix scans stream contiguous rows.  g = global slot, f = field, nf = fields per slot.
AUGX_HD int64_t fidx(int64_t g, int f, int nf) { return ((g / CHUNK) * nf + f) * CHUNK + (g % CHUNK); }
AUGX_HD int64_t listOff(const BatchView &B, int p) { return B.listOffs[p]; }
AUGX_HD int64_t pathOff(const BatchView &B, int p) { return B.off[p] / 8 + 64 * (int64_t)p; }
AUGX_HD int64_t pathCap(const BatchView &B, int p) { return (B.off[p + 1] - B.off[p]) / 8 + 64; }
// offset that turns a stored Viterbi value of base q of piece p into the true one (0 unless the piece was decoded in segments)
AUGX_HD double frameOff(const BatchView &B, int p, int q) {
    const int s0 = B.pieceSeg0[p], s1 = B.pieceSeg0[p + 1];
    if (s1 - s0 <= 1) return 0.0;
    int r = s0;
    while (r + 1 < s1 && q > B.brkPos[r]) r++;
    return B.brkOff[r];
}

// view of one piece: pointers pre-offset so that index q is the 0-based base position
// the codon c0 c1 c2 (0..3 = acgt) ends a reading frame under the translation table: mask bit 0 taa, bit 1 tag, bit 2 tga
AUGX_HD bool stopCodon3(int c0, int c1, int c2, int mask) {
    if (c0 != 3) return false;
    if (c1 == 0) return (c2 == 0 && (mask & 1)) || (c2 == 2 && (mask & 2));
    return c1 == 2 && c2 == 0 && (mask & 4);
}
struct Piece {
    const DevTables *t;
    int n, c;                  // length, GC class
    int64_t o;                 // slot offset of the piece: base q lives in slot o+1+q
    const uint8_t *code;       // code[q]
    const uint64_t *fx;        // fx[fidx(o+1+q, f, NFX)] = prefix sum up to and including q
    const uint32_t *nsm;       // nsm[fidx(o+1+q, r, 6)]
    const double *sig;         // sig[q*NSIG + i]
    // optional LDS windows (trellis kernel only): recent bases / stop tables are served from LDS, the rest from HBM
    const uint8_t *wcode = nullptr;  // wcode[q & (CODE_WIN-1)] for q in [wcLo, wcHi)
    int wcLo = 0, wcHi = 0;
    const uint32_t *wns = nullptr;   // wns[(q & (NS_WIN-1))*6 + r] for q in [wnLo, wnHi)
    int wnLo = 0, wnHi = 0;
    const uint8_t *lcode = nullptr;  // (prep kernels) the bases lLo..lHi-1 staged by the workgroup: lcode[q - lLo], 4 outside the piece
    int lLo = 0, lHi = 0;
    AUGX_HD int b(int p) const {
#if defined(__HIP_DEVICE_COMPILE__)
        // (two typed loads behind a branch; one load through a pointer selected between LDS and HBM would be a flat load, which
        //  holds up the LDS and the memory wait counters alike)
        if (p >= lLo && p < lHi) return *(const __attribute__((address_space(3))) uint8_t *)(lcode + (p - lLo));
        return (p >= 0 && p < n) ? *(const __attribute__((address_space(1))) uint8_t *)(code + p) : 4;
#else
        if (p >= lLo && p < lHi) return lcode[p - lLo];
        return (p >= 0 && p < n) ? code[p] : 4;
#endif
    }
    AUGX_HD bool is2(int p, int x, int y) const { return b(p) == x && b(p + 1) == y; }
    AUGX_HD int pat(int p, int len) const {
        int r = 0;
        for (int i = 0; i < len; i++) {
            int cc = b(p + i);
            if (cc > 3) return -1;
            r = (r << 2) | cc;
        }
        return r;
    }
    AUGX_HD int rcpat(int p, int len) const { // Seq2Int::rc, reference include/geneticcode.hh:174-179
        int r = 0;
        for (int i = 0; i < len; i++) {
            int cc = b(p + i);
            if (cc > 3) return -1;
            r |= (3 - cc) << (2 * i);
        }
        return r;
    }
    // (which of taa / tag / tga end a reading frame: the translation table, DevTables::stopMask)
    AUGX_HD bool isStop(int p) const { return stopCodon3(b(p), b(p + 1), b(p + 2), t->stopMask); }
    AUGX_HD bool isRCStop(int p) const {
        const int c0 = b(p + 2), c1 = b(p + 1), c2 = b(p);
        return stopCodon3(c0 <= 3 ? 3 - c0 : 4, c1 <= 3 ? 3 - c1 : 4, c2 <= 3 ? 3 - c2 : 4, t->stopMask);
    }
    // splice-site gates, reference include/statemodel.hh:98-117 (ab initio: consensus dinucleotides only)
    // (onGenDSS / onGenRDSS, include/geneticcode.hh:47-54: gt -- reverse strand: ac -- or, where the species allows it, gc)
    AUGX_HD bool possDSS(int pos) const { return pos >= 1 && pos <= n - 2 && (is2(pos, 2, 3) || (t->dssGc && is2(pos, 2, 1))); }
    AUGX_HD bool possRDSS(int pos) const { return pos >= 1 && pos <= n - 2 && (is2(pos - 1, 0, 1) || (t->dssGc && is2(pos - 1, 2, 1))); }
    AUGX_HD bool possASS(int pos) const { return pos >= 1 && pos <= n - 2 && is2(pos - 1, 0, 2); }
    AUGX_HD bool possRASS(int pos) const { return pos >= 1 && pos <= n - 2 && is2(pos, 1, 3); }
    // Markov-chain content of bases l..r from the fixed-point prefix field f
    AUGX_HD double seg(int f, int l, int r) const {
        if (l > r) return 0.0;
        uint64_t hi = fx[fidx(o + 1 + r, f, NFX)], lo = fx[fidx(o + l, f, NFX)];
        return (double)(int64_t)(hi - lo) * AUGX_FX_INV;
    }
    // nearest in-frame stop: reference OpenReadingFrame tables, src/exonmodel.cc:101-156
    AUGX_HD int nearestStop(int pos, bool fwd) const {
        if (n <= 5 && pos >= n - 2) return 0; // tables left unpatched by the reference for tiny inputs
        if (pos >= wnLo && pos < wnHi) return (int)wns[(pos & (NS_WIN - 1)) * 6 + (fwd ? 0 : 3) + pos % 3] - 1;
        return (int)nsm[fidx(o + 1 + pos, (fwd ? 0 : 3) + pos % 3, 6)] - 1;
    }
    // reference OpenReadingFrame::leftmostExonBegin, src/exonmodel.cc:165-198
    AUGX_HD int leftmostExonBegin(int frame, int base, bool fwd) const {
        int pos;
        if (fwd) pos = (frame == 0 || frame == 1) ? base - frame - 3 : base - frame;
        else pos = (frame == 1 || frame == 2) ? base + frame - 5 : base - 2;
        if (pos >= n) pos -= 3 * ((pos - n + 3) / 3);
        int lmb = pos >= 0 ? nearestStop(pos, fwd) + 1 : 0;
        int maxAllowed = t->max_exon_len - t->U - t->As - 2 - 2 - t->Ds;
        if (lmb < base - maxAllowed) lmb = base - maxAllowed;
        return lmb;
    }
};

AUGX_HD uint64_t toFx(double lnp) { return (uint64_t)(int64_t)llrint(lnp * AUGX_FX_SCALE); }

// Motif::seqProb (reference src/motif.cc:308-331), forward and reverse-complement
AUGX_HD double motifF(const Piece &P, const double *m, int mn, int mk, int start) {
    double s = 0;
    int sz = 1 << (2 * (mk + 1));
    for (int i = 0; i < mn; i++) {
        int pn = P.pat(start + i - mk, mk + 1);
        s += pn >= 0 ? m[(int64_t)i * sz + pn] : P.t->ln_quarter;
    }
    return s;
}
AUGX_HD double motifRC(const Piece &P, const double *m, int mn, int mk, int start) {
    double s = 0;
    int sz = 1 << (2 * (mk + 1));
    for (int i = 0; i < mn; i++) {
        int pn = P.rcpat(start + i, mk + 1);
        s += pn >= 0 ? m[(int64_t)(mn - 1 - i) * sz + pn] : P.t->ln_quarter;
    }
    return s;
}
AUGX_HD double tisBin(const DevTables &t, int c, double lnp) {
    if (t.tis_nbins < 1) return lnp;
    double p = exp(lnp);
    const double *bb = t.tis_bin_bounds + (int64_t)c * (t.tis_nbins - 1);
    int a = 0, bq = t.tis_nbins - 1;
    while (a < bq) {
        int m = (a + bq) / 2;
        if (p < bb[m]) bq = m; else a = m + 1;
    }
    return t.tis_bin_ln[(int64_t)c * t.tis_nbins + a];
}
// IntronModel::dSSProb, reference src/intronmodel.cc:1195-1248
AUGX_HD double dssProb(const Piece &P, int base, bool fwd) {
    const DevTables &t = *P.t;
    int a = 0, bq = 0;
    bool nonGt;
    if (fwd) {
        int dsspos = base + t.Ds;
        if (!P.possDSS(dsspos)) return AUGX_NINF;
        nonGt = !P.is2(dsspos, 2, 3);
        a = P.pat(base, t.Ds);
        bq = P.pat(dsspos + 2, t.De);
        if (a < 0 || bq < 0) return AUGX_NINF;
    } else {
        int dsspos = base + t.De;
        if (!P.possRDSS(dsspos + 1)) return AUGX_NINF;
        nonGt = !P.is2(dsspos, 0, 1);
        for (int i = 0; i < t.Ds; i++) { int cc = P.b(dsspos + 2 + t.Ds - 1 - i); if (cc > 3) return AUGX_NINF; a = (a << 2) | (3 - cc); }
        for (int i = 0; i < t.De; i++) { int cc = P.b(base + t.De - 1 - i); if (cc > 3) return AUGX_NINF; bq = (bq << 2) | (3 - cc); }
    }
    // (a gc site: the second half of the table -- pattern probability times non_gt_dss_prob, binned; src/intronmodel.cc:1232-1239)
    return t.dss_pat[((a << (2 * t.De)) | bq) + ((nonGt && t.dssGc) ? (1 << (2 * (t.Ds + t.De))) : 0)];
}
// IntronModel::aSSProb, reference src/intronmodel.cc:1116-1188
AUGX_HD double assProb(const Piece &P, int base, bool fwd) {
    const DevTables &t = *P.t;
    const double *M = t.ass_motif + (int64_t)P.c * t.ass_n * (1 << (2 * (t.ass_k + 1)));
    double motif;
    int a = 0, bq = 0;
    bool valid = true;
    if (fwd) {
        int asspos = base + t.U + t.As;
        if (!P.possASS(asspos + 1)) return AUGX_NINF;
        for (int i = 0; i < t.As; i++) { int cc = P.b(base + t.U + i); if (cc > 3) valid = false; a = (a << 2) | (cc & 3); }
        for (int i = 0; i < t.Ae; i++) { int cc = P.b(asspos + 2 + i); if (cc > 3) valid = false; bq = (bq << 2) | (cc & 3); }
        motif = base >= t.ass_k ? motifF(P, M, t.ass_n, t.ass_k, base) : AUGX_NINF;
    } else {
        int asspos = base + t.Ae;
        if (!P.possRASS(asspos)) return AUGX_NINF;
        for (int i = 0; i < t.As; i++) { int cc = P.b(asspos + 2 + t.As - 1 - i); if (cc > 3) valid = false; a = (a << 2) | ((3 - cc) & 3); }
        for (int i = 0; i < t.Ae; i++) { int cc = P.b(base + t.Ae - 1 - i); if (cc > 3) valid = false; bq = (bq << 2) | ((3 - cc) & 3); }
        int motifstart = base + t.As + 2 + t.Ae, motifend = motifstart + t.U;
        motif = motifend + t.ass_k < P.n ? motifRC(P, M, t.ass_n, t.ass_k, motifstart) : t.U * t.ln_quarter;
    }
    double patl = valid ? t.ass_pat[(a << (2 * t.Ae)) | bq] : t.ass_pat_invalid;
    return motif + patl;
}
// begin part of SINGLE/INITIAL exons as a function of the start codon position bob
// (reference ExonModel::notEndPartEmiProb, src/exonmodel.cc:1427-1463)
AUGX_HD double tisFwd(const Piece &P, int bob) {
    const DevTables &t = *P.t;
    if (!(bob >= 0 && bob < P.n - 2)) return AUGX_NINF;
    int pn = P.pat(bob, 3);
    if (pn < 0) return AUGX_NINF;
    double begin = t.ln_startcodon[pn];
    if (begin == AUGX_NINF) return AUGX_NINF;
    int tis = bob - t.W;
    if (tis > t.tis_k) {
        const double *M = t.tis_motif + (int64_t)P.c * t.tis_n * (1 << (2 * (t.tis_k + 1)));
        return tisBin(t, P.c, begin + motifF(P, M, t.tis_n, t.tis_k, tis));
    }
    return begin + bob * t.ln_quarter; // pow(0.25, beginOfStart - STARTCODON_LEN)
}
// end part of RSINGLE/RINITIAL exons ending (as a state) at `end` (reference endPartEmiProb, :1313-1350)
AUGX_HD double tisRev(const Piece &P, int end) {
    const DevTables &t = *P.t;
    int startpos = end - t.W - 3 + 1;
    if (startpos < 0) return AUGX_NINF;
    int pn = P.rcpat(startpos, 3);
    if (pn < 0 || t.ln_startcodon[pn] == AUGX_NINF) return AUGX_NINF;
    if (startpos + 3 + t.W - 1 + t.tis_mem < P.n) {
        const double *M = t.tis_motif + (int64_t)P.c * t.tis_n * (1 << (2 * (t.tis_k + 1)));
        return tisBin(t, P.c, t.ln_startcodon[pn] + motifRC(P, M, t.tis_n, t.tis_k, startpos + 3));
    }
    return (P.n - (startpos + 3)) * t.ln_quarter;
}
// single-base emissions (reference src/igenicmodel.cc:328-356, src/intronmodel.cc:895-915)
AUGX_HD double eIg(const Piece &P, int p) {
    const DevTables &t = *P.t;
    if (p > t.k) {
        int pn = P.pat(p - t.k, t.k + 1);
        return pn >= 0 ? t.ig_emi[(int64_t)P.c * t.NP + pn] : t.ln_quarter;
    }
    int bk = P.pat(0, p + 1);
    return bk >= 0 ? t.ig_short[((int64_t)P.c * (t.k + 1) + p) * t.NP + bk] : t.ln_quarter;
}
AUGX_HD double eIn(const Piece &P, int p) {
    const DevTables &t = *P.t;
    int pn = p >= t.kIn ? P.pat(p - t.kIn, t.kIn + 1) : -1;
    return pn >= 0 ? t.in_emi[(int64_t)P.c * t.NPin + pn] : t.ln_quarter;
}
// a base of a utr5intron / utr3intron state where the UTR order uk is below k: the intron pattern read from p - uk on, k + 1 bases
// -- it ends k - uk bases AFTER p; past the piece: 1/4 (reference src/utrmodel.cc:1255-1262: s2i_intron(sequence + pos - k) with UtrModel::k)
AUGX_HD double eUin(const Piece &P, int p) {
    const DevTables &t = *P.t;
    int pn = (p >= t.uk && p - t.uk + t.kIn < P.n) ? P.pat(p - t.uk, t.kIn + 1) : -1;
    return pn >= 0 ? t.in_emi[(int64_t)P.c * t.NPin + pn] : t.ln_quarter;
}

struct ExGeom { int bpl, ipo, baseOffset, ipeo; bool fwd; };
AUGX_HD ExGeom exGeom(const DevTables &t, int kind) { // reference src/exonmodel.cc:231-279
    ExGeom g;
    g.fwd = kind == AUGX_K_SINGLE || kind == AUGX_K_INITIAL || kind == AUGX_K_INTERNAL || kind == AUGX_K_TERMINAL;
    if (kind == AUGX_K_SINGLE || kind == AUGX_K_INITIAL) { g.bpl = 3 + t.W; g.ipo = 3; }
    else if (kind == AUGX_K_RSINGLE || kind == AUGX_K_RTERMINAL) { g.bpl = g.ipo = 3; }
    else { g.bpl = 0; g.ipo = g.fwd ? t.Ae : t.Ds; }
    if (kind == AUGX_K_SINGLE || kind == AUGX_K_TERMINAL) { g.baseOffset = 0; g.ipeo = 3; }
    else if (kind == AUGX_K_RSINGLE || kind == AUGX_K_RINITIAL) { g.baseOffset = -t.W; g.ipeo = 3; }
    else { g.baseOffset = g.ipeo = g.fwd ? t.Ds : t.Ae; }
    return g;
}
// reference ExonModel::endPartEmiProb, src/exonmodel.cc:1272-1400.  tisR = precomputed tisRev(end)
AUGX_HD double exEndPart(const Piece &P, int kind, int win, int end, double tisR) {
    const DevTables &t = *P.t;
    const int n = P.n;
    switch (kind) {
    case AUGX_K_SINGLE: case AUGX_K_TERMINAL: {
        int stp = end - 2;
        if (stp < 0 || stp > n - 3 || !P.isStop(stp)) return AUGX_NINF;
        if (P.b(stp + 1) == 0 && P.b(stp + 2) == 0) return t.ln_stop_ochre;
        if (P.b(stp + 1) == 0 && P.b(stp + 2) == 2) return t.ln_stop_amber;
        return t.ln_stop_opal;
    }
    case AUGX_K_RSINGLE: case AUGX_K_RINITIAL:
        return tisR;
    case AUGX_K_INITIAL: case AUGX_K_INTERNAL: {
        int dsspos = end + t.Ds + 1;
        if (end == n - 1) return 0.0;
        if ((dsspos + 2 - 1 < n && !P.possDSS(dsspos)) || end + t.Ds >= n || P.leftmostExonBegin(win - 1, end + t.Ds, true) >= end)
            return AUGX_NINF;
        return 0.0;
    }
    default: {
        int asspos = end + t.Ae + 1;
        if (end == n - 1) return 0.0;
        if (end + t.Ae + 2 < n && P.possRASS(asspos)) return 0.0;
        return AUGX_NINF;
    }
    }
}
// reference ExonModel::notEndPartEmiProb, src/exonmodel.cc:1417-1859 (ab initio: extrinsicQuot == 1).
// tisF = value of sig[SIG_TISF] at bob (only read for SINGLE/INITIAL)
AUGX_HD double exNotEndPart(const Piece &P, int kind, int win, int bs, int right, int fOR, const ExGeom &g, double tisF) {
    const DevTables &t = *P.t;
    const int n = P.n, k = t.k, c = P.c;
    (void)n;
    double begin;
    int bob = bs - g.ipo;
    switch (kind) {
    case AUGX_K_SINGLE: case AUGX_K_INITIAL:
        begin = tisF;
        if (begin == AUGX_NINF) return AUGX_NINF;
        break;
    case AUGX_K_TERMINAL: case AUGX_K_INTERNAL:
        if (bs > 0) {
            if (bob < 0 || (bob - 2 >= 0 && !P.possASS(bob - 1))) return AUGX_NINF;
            begin = 0.0;
        } else if (bs == 0) begin = 0.0;
        else return AUGX_NINF;
        break;
    case AUGX_K_RSINGLE: case AUGX_K_RTERMINAL:
        if (bob < 0) return AUGX_NINF;
        if (P.b(bob) == 3 && P.b(bob + 1) == 3 && P.b(bob + 2) == 0) begin = t.ln_stop_ochre;
        else if (P.b(bob) == 1 && P.b(bob + 1) == 3 && P.b(bob + 2) == 0) begin = t.ln_stop_amber;
        else if (P.b(bob) == 3 && P.b(bob + 1) == 1 && P.b(bob + 2) == 0) begin = t.ln_stop_opal;
        else return AUGX_NINF;
        if (begin == AUGX_NINF) return AUGX_NINF;
        break;
    default:
        if (bs == 0) begin = 0.0;
        else if (bob < 0 || (bob - 2 > 0 && !P.possRDSS(bob - 1))) return AUGX_NINF;
        else begin = 0.0;
    }
    double rest;
    if (bs > right) {
        rest = (bs - right - 1) * t.ln4;
    } else if (right - bs <= k) {
        int l = right - bs;
        int pn = g.fwd ? P.pat(bs, l + 1) : P.rcpat(bs, l + 1);
        if (pn >= 0) {
            int f = g.fwd ? fOR : mod3(fOR + right - bs);
            rest = AUGX_GTAB(t.ex_pls)[(((int64_t)c * (k + 1) + l) * 3 + f) * t.NP + pn];
        } else
            rest = (l + 1) * t.ln_n_coding;
    } else {
        int endOfStart = bs + k - 1, beginOfInitP = right - (k - 1);
        if (k == 0) rest = 0;
        else if (g.fwd) {
            int pn = P.pat(bs, k);
            rest = pn >= 0 ? AUGX_GTAB(t.ex_pls)[(((int64_t)c * (k + 1) + (k - 1)) * 3 + mod3(fOR - right + endOfStart)) * t.NP + pn] : k * t.ln_n_coding;
        } else {
            int pn = P.rcpat(beginOfInitP, k);
            rest = pn >= 0 ? AUGX_GTAB(t.ex_pls)[(((int64_t)c * (k + 1) + (k - 1)) * 3 + mod3(fOR + right - beginOfInitP)) * t.NP + pn] : k * t.ln_n_coding;
        }
        const int a = g.fwd ? mod3(fOR - right) : mod3(fOR + right);
        const int fb = ((g.fwd ? 0 : 1) * 3 + a) * 3; // field base: +0 exon, +1 init, +2 et
        const int PX = fb, PI = fb + 1, PT = fb + 2;
        int endOfInitial, beginOfTerm, endOfTerm, beginOfInitial;
        double inner;
        switch (kind) {
        case AUGX_K_SINGLE:
            endOfInitial = endOfStart + t.Li;
            if (endOfInitial > right) endOfInitial = right;
            inner = P.seg(PI, endOfStart + 1, endOfInitial) + P.seg(PX, endOfInitial + 1, right);
            break;
        case AUGX_K_INITIAL:
            endOfInitial = endOfStart + t.Li;
            if (endOfInitial > right) { endOfInitial = right; beginOfTerm = right + 1; }
            else { beginOfTerm = right - t.Le + 1; if (beginOfTerm <= endOfInitial) beginOfTerm = right + 1; }
            inner = (P.seg(PI, endOfStart + 1, endOfInitial) + P.seg(PX, endOfInitial + 1, beginOfTerm - 1)) + P.seg(PT, beginOfTerm, right);
            break;
        case AUGX_K_INTERNAL:
            beginOfTerm = right - t.Le + 1;
            if (beginOfTerm <= endOfStart) beginOfTerm = right + 1;
            inner = P.seg(PX, endOfStart + 1, beginOfTerm - 1) + P.seg(PT, beginOfTerm, right);
            break;
        case AUGX_K_TERMINAL:
            inner = P.seg(PX, endOfStart + 1, right);
            break;
        case AUGX_K_RSINGLE:
            beginOfInitial = beginOfInitP - t.Li;
            if (beginOfInitial < bs) beginOfInitial = bs;
            inner = P.seg(PI, beginOfInitial, beginOfInitP - 1) + P.seg(PX, bs, beginOfInitial - 1);
            break;
        case AUGX_K_RINITIAL:
            beginOfInitial = beginOfInitP - t.Li;
            if (beginOfInitial < bs) { beginOfInitial = bs; endOfTerm = bs - 1; }
            else { endOfTerm = bs + t.Le - 1; if (endOfTerm >= beginOfInitial) endOfTerm = bs - 1; }
            inner = (P.seg(PI, beginOfInitial, beginOfInitP - 1) + P.seg(PX, endOfTerm + 1, beginOfInitial - 1)) + P.seg(PT, bs, endOfTerm);
            break;
        case AUGX_K_RINTERNAL:
            endOfTerm = bs + t.Le - 1;
            if (endOfTerm >= beginOfInitP) endOfTerm = bs - 1;
            inner = P.seg(PX, endOfTerm + 1, beginOfInitP - 1) + P.seg(PT, bs, endOfTerm);
            break;
        default:
            inner = P.seg(PX, bs, beginOfInitP - 1);
        }
        rest = rest + inner;
    }
    int eob = right + g.ipeo;
    int len = eob - bob + 1;
    double lenPart;
    if (len < 1 || len > t.max_exon_len) return AUGX_NINF;
    switch (kind) {
    case AUGX_K_SINGLE: case AUGX_K_RSINGLE: lenPart = len % 3 == 0 ? AUGX_GTAB(t.len_single)[len] : AUGX_NINF; break;
    case AUGX_K_INITIAL: lenPart = (len % 3 == win && len > 2) ? AUGX_GTAB(t.len_initial)[len] : AUGX_NINF; break;
    case AUGX_K_RINITIAL: lenPart = len > 2 ? AUGX_GTAB(t.len_initial)[len] : AUGX_NINF; break;
    case AUGX_K_INTERNAL: case AUGX_K_RINTERNAL: lenPart = AUGX_GTAB(t.len_internal)[len]; break;
    case AUGX_K_TERMINAL: lenPart = AUGX_GTAB(t.len_terminal)[len]; break;
    default: lenPart = mod3(2 - len) == win ? AUGX_GTAB(t.len_terminal)[len] : AUGX_NINF;
    }
    if (lenPart == AUGX_NINF) return AUGX_NINF;
    return (begin + rest) + lenPart;
}

// geometry of one exon end (state s ending at base j), reference src/exonmodel.cc:939-1054
struct ExEnd { int eob, right, fOR, startMin, startMax; };
AUGX_HD ExEnd exEnd(const Piece &P, int kind, int win, int j, const ExGeom &g) {
    const DevTables &t = *P.t;
    ExEnd e;
    e.eob = j + g.baseOffset;
    e.right = e.eob - g.ipeo;
    e.fOR = g.fwd ? mod3(win - (e.eob + 1) + e.right) : mod3(win + e.eob + 1 - e.right);
    int eons = (kind == AUGX_K_TERMINAL || kind == AUGX_K_SINGLE) ? e.eob - 3 : e.eob;
    if (eons > P.n - 1) eons = P.n - 1;
    int feons = g.fwd ? mod3(win - 1 - e.eob + eons) : mod3(win + 1 + e.eob - eons);
    int ORFleft = P.leftmostExonBegin(feons, eons, g.fwd);
    e.startMax = e.eob + g.ipo - t.min_exon_len + 1;
    if (kind == AUGX_K_RTERMINAL || kind == AUGX_K_RSINGLE)
        e.startMin = e.startMax = ORFleft + 2;
    else {
        e.startMin = ORFleft <= 0 ? 0 : ORFleft + g.ipo;
        if (e.startMax > j + g.bpl) e.startMax = j + g.bpl;
    }
    return e;
}

// lessD / rlessD end gate, reference src/intronmodel.cc:545-556
AUGX_HD bool lessDGate(const Piece &P, bool fwd, int j) {
    const DevTables &t = *P.t;
    int eobi = fwd ? j + t.U + t.As + 2 : j + t.De + 2;
    if (eobi - 2 + 1 < P.n - 1) return fwd ? P.possASS(eobi) : P.possRDSS(eobi);
    return true;
}

// ---- exon-like UTR states (reference UtrModel::viterbiForwardAndSampling, src/utrmodel.cc)
// window [lm, rm] of predecessor ends (:822-916, with the clamps of :941-952)
AUGX_HD void utrWindow(const DevTables &T, int kind, int j, int n, int &lm, int &rm) {
    const int W = T.W, U = T.U, up = T.tss_upwin, te = T.tss_end, dc = T.dpc, bl = T.boxlen, assWhole = T.As + 2 + T.Ae, dssWhole = T.Ds + 2 + T.De;
    const int ML = T.uML, M3S = T.uM3S, M3T = T.uM3T;
    switch (kind) {
    case AUGX_K_UTR5SINGLE: lm = j - (ML - W + up); rm = j - up - te - 1 + W + te; if (rm > j - 1) rm = j - 1; break;
    case AUGX_K_RUTR5SINGLE: lm = j - (ML - W + up); rm = j - up - 1 + W; if (rm > j - 1) rm = j - 1; break;
    case AUGX_K_UTR5INIT: case AUGX_K_RUTR5INIT: lm = j - (ML + 2 + T.De + up); rm = j - up - te - dssWhole; break;
    case AUGX_K_UTR5INTERNAL: case AUGX_K_RUTR5INTERNAL: case AUGX_K_UTR3INTERNAL: case AUGX_K_RUTR3INTERNAL:
        lm = j - (ML + 2 + T.De + U + T.As + 2); rm = j - dssWhole - U - assWhole; break;
    case AUGX_K_UTR5TERM: case AUGX_K_RUTR5TERM:
        lm = j - (ML - W + U + T.As + 2); rm = j - U - assWhole;
        if (-U - assWhole + W + T.Ae < 0) rm = j - U - assWhole + W + T.Ae;
        break;
    case AUGX_K_UTR3SINGLE: lm = j - M3S; rm = j != n - 1 ? j - dc - bl : j - 1; break;
    case AUGX_K_RUTR3SINGLE: lm = j - M3S; rm = j - dc - bl; break;
    case AUGX_K_UTR3INIT: case AUGX_K_RUTR3INIT: lm = j - (ML + 2 + T.De); rm = j - T.De - 2; break;
    case AUGX_K_UTR3TERM: lm = j - (M3T + 2 + T.As + U); rm = j != n - 1 ? j - dc - bl - assWhole - U : j - assWhole - U; break;
    default: lm = j - (M3T + 2 + T.As + U); rm = j - dc - bl - assWhole - U;
    }
    if (kind == AUGX_K_UTR5SINGLE || kind == AUGX_K_UTR5INIT) { if (lm < -up) lm = -up; }
    else if (kind == AUGX_K_RUTR3SINGLE || kind == AUGX_K_RUTR3TERM) { if (lm < -bl - dc) lm = -bl - dc; }
    else if (lm < 0) lm = 0;
}
// smallest distance j - rm of any UTR exon kind: the block size of the dense kernels must not exceed it
AUGX_HD int utrMinLag(const DevTables &T) {
    int m = 1 << 30;
    for (int kind = AUGX_K_UTR5SINGLE; kind <= AUGX_K_RUTR3TERM; kind++) {
        if (!isUtrExonKind(kind)) continue;
        int lm, rm;
        utrWindow(T, kind, 100000, 1 << 29, lm, rm);
        if (100000 - rm < m) m = 100000 - rm;
    }
    return m;
}

} // namespace dev
} // namespace augx
