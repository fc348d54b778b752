// dense.h -- the "dense" decode kernels: models whose state graph the wavefront layout of trellisPiece (kernels.h) was not built
// for -- first of all the 71-state model with untranslated regions (--UTR=on, reference src/utrmodel.cc).
//
// What differs from the 47-state path:
//   * the ln V matrix is kept dense in HBM ([N][S], BatchView::cells; ln F in BatchView::fwd): a candidate names its predecessor
//     by (end of predecessor, state), no value lists; the newest 64 columns also live in an LDS ring;
//   * one kernel body, densePiece<BLK, MODE>, does the Viterbi recurrence (MODE 0: max, bit-exact) and the forward recurrence
//     (MODE 1: ln-sum), data-driven by the state graph: fixed-lag states, early chain states (geometric introns), candidates of
//     the variable-length states, late chain states (intergenic, UTR introns), reverse terminal exons -- in that order per block;
//   * the candidates of the 16 exon-like UTR states are NOT materialised (they would be ~200 records per base): a UTR exon that
//     ends at base j in state s is scored  V[eop][a] + t(a->s) + B_s[eop] + E_s[j] + lenDist_s[j - eop + c_s]  where B (begin
//     signal minus content prefix, per begin SITE, written once by the prep kernels into short site lists) and E (end signal plus
//     content prefix, per end base) do not depend on V -- the trellis walks the site list of the state's window
//     (reference UtrModel::viterbiForwardAndSampling, src/utrmodel.cc:796-1064; notEndPartEmiProb :1167-1548).
// Everything is compiled for gfx950 and, with -DAUGX_EMU, for the lane-loop emulator (tests only).
#pragma once
#include "kernels.h"

namespace augx {
namespace dev {

AUGX_HD int baseClass(const BatchView &B, int p, int64_t g) { return B.cls[p] < 0 ? -1 : B.planeCls[p * MAXPL + B.gcPlane[g]]; }
AUGX_HD double fxD(uint64_t v) { return (double)(int64_t)v * AUGX_FX_INV; }

// =================================================================================================
// K1 (UTR): content prefix terms and begin-site counts, one scan; signal records, end gates and the site lists
// =================================================================================================
// terms of the NUFX content prefix fields and the NUCNT site counts of slot g.  The term of a base comes from the class OF THAT
// BASE (reference SegProbs::setEmiProbs is re-run over the region of every class, src/statemodel.cc:398-432, src/utrmodel.cc:784-790);
// base 0 counts ln 1/4 (cumProds[0] = .25)
AUGX_HD void k1UtrTermsCalc(const DevTables &T, const BatchView &B, int64_t g, uint64_t out[NUFX + NUCNT], const uint8_t *lcode = nullptr,
                            int lLo = 0, int lHi = 0) {
    const int p = B.chunkPiece[g / CHUNK];
    const int64_t o = B.off[p];
    const int q = (int)(g - o - 1), n = B.len[p];
    for (int i = 0; i < NUFX + NUCNT; i++) out[i] = 0;
    if (q < 0 || q >= n || B.cls[p] < 0 || !T.utr) return; // (a dense model without UTR states: nothing to count)
    Piece P;
    P.t = &T; P.n = n; P.c = baseClass(B, p, g); P.o = o; P.code = B.code + o + 1; P.fx = nullptr; P.nsm = nullptr; P.sig = nullptr;
    P.lcode = lcode; P.lLo = lLo; P.lHi = lHi;
    const int k = T.uk, NP = T.uNP, c = P.c;
    const int pn = (q >= 1 && q >= k) ? P.pat(q - k, k + 1) : -1;
    const int rn = (q >= 1 && q < n - k) ? P.rcpat(q, k + 1) : -1;
    const double *t5i = AUGX_GTAB(T.utr5init_emi) + (int64_t)c * NP, *t5 = AUGX_GTAB(T.utr5_emi) + (int64_t)c * NP, *t3 = AUGX_GTAB(T.utr3_emi) + (int64_t)c * NP;
    out[UFX_5IF] = toFx(pn >= 0 ? t5i[pn] : T.ln_quarter); out[UFX_5IR] = toFx(rn >= 0 ? t5i[rn] : T.ln_quarter);
    out[UFX_5F] = toFx(pn >= 0 ? t5[pn] : T.ln_quarter);   out[UFX_5R] = toFx(rn >= 0 ? t5[rn] : T.ln_quarter);
    out[UFX_3F] = toFx(pn >= 0 ? t3[pn] : T.ln_quarter);   out[UFX_3R] = toFx(rn >= 0 ? t3[rn] : T.ln_quarter);
    // a transcript may start only at every tts_spacing-th base (src/utrmodel.cc:1785); the TSS window must fit the piece (:1768)
    if (q % T.tts_spacing == 0 && q + T.tss_upwin + T.tss_end - 1 < n) out[NUFX + UCNT_TF] = 1;
    // a forward single / terminal exon may end at q (stop codon q-2..q): predecessor of the forward 3' UTR
    if (q - 2 >= 0 && P.isStop(q - 2)) {
        const double sp = (P.b(q - 1) == 0 && P.b(q) == 0) ? T.ln_stop_ochre : (P.b(q - 1) == 0 && P.b(q) == 2) ? T.ln_stop_amber : T.ln_stop_opal;
        if (sp > AUGX_NINF) out[NUFX + UCNT_FS] = 1;
    }
    // the reverse poly-A box begins at q (ttsProbMinus[q] > 0, src/utrmodel.cc:1872-1903; the state begins d_polyasig_cleavage before it)
    if (q - T.dpc > 0 && q + T.boxlen - 1 < n) {
        const int bn = P.rcpat(q, T.boxlen);
        if ((bn >= 0 && AUGX_GTAB(T.aataaa)[bn] > AUGX_NINF) || q % T.tts_spacing == 0) out[NUFX + UCNT_TM] = 1;
    }
    // a reverse single / initial exon may end at q (reverse start codon at q-W-2): predecessor of the reverse 5' UTR
    if (q - T.W - 2 >= 0) {
        const int sn = P.rcpat(q - T.W - 2, 3);
        if (sn >= 0 && T.ln_startcodon[sn] > AUGX_NINF) out[NUFX + UCNT_RT] = 1;
    }
}
AUGX_HD void k1UtrTerms(const DevTables &T, const BatchView &B, int64_t g) { // (emulator: terms to memory, scanned in place)
    uint64_t v[NUFX + NUCNT];
    k1UtrTermsCalc(T, B, g, v);
    for (int i = 0; i < NUFX; i++) B.ufx[fidx(g, i, NUFX)] = v[i];
    for (int i = 0; i < NUCNT; i++) B.ucnt[fidx(g, i, NUCNT)] = (uint32_t)v[NUFX + i];
}
// longest UTR site list of piece p (next to k1ListCount: the lists share one capacity)
AUGX_HD void k1UtrListCount(const BatchView &B, int p) {
    const int64_t g = B.off[p] + B.len[p];
    uint32_t m = (uint32_t)B.listCnt[p];
    for (int f = 0; f < NUCNT; f++) { const uint32_t c = B.ucnt[fidx(g, f, NUCNT)]; m = c > m ? c : m; }
    B.listCnt[p] = (int32_t)m;
}

// UtrModel::tssupSeqProb, src/utrmodel.cc:1733-1750
AUGX_HD double tssupSeq(const Piece &P, int left, int right, bool rev) {
    const DevTables &T = *P.t;
    const int uk = T.tssup_k;
    const double *E = AUGX_GTAB(T.tssup_emi) + (int64_t)P.c * (1 << (2 * (uk + 1)));
    double s = 0;
    for (int p = right; p >= left; p--) {
        int pn = -1;
        if (!rev && p - uk >= 0) pn = P.pat(p - uk, uk + 1);
        else if (rev && p >= 0 && p + uk < P.n) pn = P.rcpat(p, uk + 1);
        s += pn >= 0 ? E[pn] : T.ln_quarter;
    }
    return s;
}
// UtrModel::tssProb, src/utrmodel.cc:1761-1833, with the tables of class P.c (ab initio: the hint factor is 1)
AUGX_HD double tssProbCalc(const Piece &P, int left, bool fwd) {
    const DevTables &T = *P.t;
    const int n = P.n, c = P.c;
    const int right = left + T.tss_upwin + T.tss_end - 1;
    if (right >= n || left < 0) return AUGX_NINF;
    if (left % T.tts_spacing != 0) return AUGX_NINF;
    const int64_t sz0 = (int64_t)1 << (2 * (T.tss_k + 1)), sz1 = (int64_t)1 << (2 * (T.tsstata_k + 1)), sz2 = (int64_t)1 << (2 * (T.tata_k + 1));
    const double *Mtss = AUGX_GTAB(T.tss_motif) + (int64_t)c * T.tss_n * sz0, *Mtt = AUGX_GTAB(T.tsstata_motif) + (int64_t)c * T.tsstata_n * sz1,
                 *Mta = AUGX_GTAB(T.tata_motif) + (int64_t)c * T.tata_n * sz2;
    const int maxpos = T.d_tss_tata_max - T.d_tss_tata_min - 1;
    if (fwd) {
        const int w0 = right - T.tss_end - T.d_tss_tata_max + 1;
        int rel = -1;
        for (int pos = 0; pos <= maxpos; pos++) // UtrModel::findTATA :271-287
            if (P.b(w0 + pos) == 3 && P.b(w0 + pos + 1) == 0 && P.b(w0 + pos + 2) == 3 && P.b(w0 + pos + 3) == 0 && P.b(w0 + pos + 5) == 0) { rel = pos; break; }
        if (rel >= 0) {
            const int tatapos = w0 + rel;
            return motifF(P, Mtt, T.tsstata_n, T.tsstata_k, right - T.tss_end - T.tss_start + 1) + motifF(P, Mta, T.tata_n, T.tata_k, tatapos - T.tata_start) +
                   (tssupSeq(P, left, tatapos - T.tata_start - 1, false) + tssupSeq(P, tatapos + T.tata_end, right - T.tss_end - T.tss_start, false));
        }
        return motifF(P, Mtss, T.tss_n, T.tss_k, right - T.tss_end - T.tss_start + 1) + tssupSeq(P, left, right - T.tss_end - T.tss_start, false);
    }
    const int w0 = left + T.tss_end + T.d_tss_tata_max - 1;
    int rel = 1;
    for (int pos = 0; pos >= -maxpos; pos--)
        if (P.b(w0 + pos) == 0 && P.b(w0 + pos - 1) == 3 && P.b(w0 + pos - 2) == 0 && P.b(w0 + pos - 3) == 3 && P.b(w0 + pos - 5) == 3) { rel = pos; break; }
    if (rel <= 0) {
        const int tatapos = w0 + rel;
        return motifRC(P, Mtt, T.tsstata_n, T.tsstata_k, left) + motifRC(P, Mta, T.tata_n, T.tata_k, tatapos - T.tata_end + 1) +
               (tssupSeq(P, left + T.tata_end + T.tata_start - 1, tatapos - T.tata_end, true) + tssupSeq(P, tatapos + T.tata_start + 1, right, true));
    }
    return motifRC(P, Mtss, T.tss_n, T.tss_k, left) + tssupSeq(P, left + T.tss_end + T.tss_start, right, true);
}
// UtrModel::computeTtsProbs, src/utrmodel.cc:1840-1912, for the box beginning at b (class P.c = the class of base b)
AUGX_HD double ttsPlusCalc(const Piece &P, int b) {
    const DevTables &T = *P.t;
    if (b < 1 || b + T.boxlen + T.dpc - 1 >= P.n) return AUGX_NINF;
    if (b - T.dpc < 0 || b + T.boxlen - 1 >= P.n) return AUGX_NINF; // (sic: the reference zeroes the PLUS entry in its minus-strand branch, :1873-1874)
    const int pn = P.pat(b, T.boxlen);
    double prob = pn >= 0 ? AUGX_GTAB(T.aataaa)[pn] : AUGX_NINF;
    if (b % T.tts_spacing == 0 && prob == AUGX_NINF) prob = T.ln_tts_rand;
    if (prob > AUGX_NINF) prob = prob + motifF(P, AUGX_GTAB(T.tts_motif) + (int64_t)P.c * T.tts_n * (1 << (2 * (T.tts_k + 1))), T.tts_n, T.tts_k, b + T.boxlen);
    return prob;
}
AUGX_HD double ttsMinusCalc(const Piece &P, int b) {
    const DevTables &T = *P.t;
    if (b < 1 || b - T.dpc < 0 || b + T.boxlen - 1 >= P.n) return AUGX_NINF;
    const int rn = P.rcpat(b, T.boxlen);
    double prob = rn >= 0 ? AUGX_GTAB(T.aataaa)[rn] : AUGX_NINF;
    if (b % T.tts_spacing == 0 && prob == AUGX_NINF) prob = T.ln_tts_rand;
    if (prob > AUGX_NINF) prob = prob + motifRC(P, AUGX_GTAB(T.tts_motif) + (int64_t)P.c * T.tts_n * (1 << (2 * (T.tts_k + 1))), T.tts_n, T.tts_k, b - T.dpc);
    return prob;
}
AUGX_HD bool ttsPlusOpen(const Piece &P, int b) { // ttsProbPlus[b] > 0 without the motif product
    const DevTables &T = *P.t;
    if (b < 1 || b + T.boxlen + T.dpc - 1 >= P.n || b - T.dpc < 0) return false;
    const int pn = P.pat(b, T.boxlen);
    return (pn >= 0 && AUGX_GTAB(T.aataaa)[pn] > AUGX_NINF) || b % T.tts_spacing == 0;
}

// geometry of the 16 exon-like UTR kinds (reference getEndPositions :1572-1643, the predecessor windows :822-916, the begin / middle
// / length parts of notEndPartEmiProb :1167-1420): which site list holds the possible begins, which of its three (signal - prefix)
// values and which content prefix field belong to the kind, first base of the middle part and of the biological exon relative to
// the begin of the state, the correction for a middle part of negative length, the length distribution
struct UGeom { int8_t list, bsel, fxf, ovl, len; int cb, cbobe; };
constexpr int UL_TF = 0, UL_LA = 1, UL_FS = 2, UL_LR = 3, UL_TM = 4, UL_RT = 5;
AUGX_HD UGeom utrGeom(const DevTables &T, int kind) {
    const int assWhole = T.As + 2 + T.Ae, dssWhole = T.Ds + 2 + T.De;
    UGeom g{0, 0, 0, 0, 0, 0, 0};
    switch (kind) {
    case AUGX_K_UTR5SINGLE: g = {UL_TF, 0, UFX_5IF, 1, 0, T.tss_upwin + T.tss_end, T.tss_upwin}; break;
    case AUGX_K_UTR5INIT: g = {UL_TF, 0, UFX_5IF, 0, 1, T.tss_upwin + T.tss_end, T.tss_upwin}; break;
    case AUGX_K_UTR5INTERNAL: g = {UL_LA, 0, UFX_5F, 0, 2, T.U + assWhole, T.U + T.As + 2}; break;
    case AUGX_K_UTR5TERM: g = {UL_LA, 0, UFX_5F, 2, 3, T.U + assWhole, T.U + T.As + 2}; break;
    case AUGX_K_UTR3SINGLE: g = {UL_FS, 0, UFX_3F, 0, 4, 0, 0}; break;
    case AUGX_K_UTR3INIT: g = {UL_FS, 0, UFX_3F, 2, 5, 0, 0}; break;
    case AUGX_K_UTR3INTERNAL: g = {UL_LA, 1, UFX_3F, 0, 6, T.U + assWhole, T.U + T.As + 2}; break;
    case AUGX_K_UTR3TERM: g = {UL_LA, 1, UFX_3F, 0, 7, T.U + assWhole, T.U + T.As + 2}; break;
    case AUGX_K_RUTR5SINGLE: g = {UL_RT, 0, UFX_5IR, 1, 0, 0, -T.W}; break;
    case AUGX_K_RUTR5INIT: g = {UL_LR, 1, UFX_5IR, 0, 1, dssWhole, T.De + 2}; break;
    case AUGX_K_RUTR5INTERNAL: g = {UL_LR, 0, UFX_5R, 0, 2, dssWhole, T.De + 2}; break;
    case AUGX_K_RUTR5TERM: g = {UL_RT, 1, UFX_5R, 2, 3, 0, -T.W}; break;
    case AUGX_K_RUTR3SINGLE: g = {UL_TM, 0, UFX_3R, 0, 4, T.boxlen + T.dpc, 0}; break;
    case AUGX_K_RUTR3INIT: g = {UL_LR, 2, UFX_3R, 2, 5, dssWhole, T.De + 2}; break;
    case AUGX_K_RUTR3INTERNAL: g = {UL_LR, 2, UFX_3R, 0, 6, dssWhole, T.De + 2}; break;
    default: g = {UL_TM, 0, UFX_3R, 0, 7, T.boxlen + T.dpc, 0}; // RUTR3TERM
    }
    return g;
}
// end of the state at j: first base of the end signal and last base of the biological exon (getEndPositions)
AUGX_HD void utrEndPos(const DevTables &T, int kind, int j, int n, int &boep, int &eobe) {
    const int assWhole = T.As + 2 + T.Ae, dssWhole = T.Ds + 2 + T.De;
    switch (kind) {
    case AUGX_K_UTR5SINGLE: case AUGX_K_UTR5TERM: boep = j + 1; eobe = j + T.W; break;
    case AUGX_K_RUTR5SINGLE: case AUGX_K_RUTR5INIT: boep = j - T.tss_upwin - T.tss_end + 1; eobe = j - T.tss_upwin; break;
    case AUGX_K_UTR5INIT: case AUGX_K_UTR5INTERNAL: case AUGX_K_UTR3INIT: case AUGX_K_UTR3INTERNAL: boep = j - dssWhole + 1; eobe = j - T.De - 2; break;
    case AUGX_K_RUTR5INTERNAL: case AUGX_K_RUTR5TERM: case AUGX_K_RUTR3INTERNAL: case AUGX_K_RUTR3TERM: boep = j - assWhole - T.U + 1; eobe = j - T.U - T.As - 2; break;
    case AUGX_K_RUTR3SINGLE: case AUGX_K_RUTR3INIT: boep = j + 1; eobe = j; break;
    default: if (j != n - 1) { boep = j - T.dpc - T.boxlen + 1; eobe = j; } else { boep = n; eobe = n - 1; }
    }
}
// does the UTR exon state of this kind pass its end gate at base j?  (endPartEmiProb > 0, :1072-1115; values: utrDescribe)
AUGX_HD bool utrGateOpen(const Piece &P, int kind, int j, double tssR) {
    const DevTables &T = *P.t;
    const int n = P.n;
    int boep, eobe;
    utrEndPos(T, kind, j, n, boep, eobe);
    if (boep < 0) return false;
    switch (kind) {
    case AUGX_K_UTR5SINGLE: case AUGX_K_UTR5TERM: {
        if (eobe + 3 > n - 1) return true;
        const int pn = P.pat(eobe + 1, 3);
        return pn >= 0 && ((T.startMask >> pn) & 1ull); // GeneticCode::isStartcodon: the translation table's start codons ({a,c,t}tg in table 1)
    }
    case AUGX_K_UTR5INIT: case AUGX_K_UTR5INTERNAL: case AUGX_K_UTR3INIT: case AUGX_K_UTR3INTERNAL: return dssProb(P, boep, true) > AUGX_NINF;
    case AUGX_K_RUTR5INTERNAL: case AUGX_K_RUTR5TERM: case AUGX_K_RUTR3INTERNAL: case AUGX_K_RUTR3TERM: return P.possRASS(boep + T.Ae);
    case AUGX_K_RUTR5SINGLE: case AUGX_K_RUTR5INIT: return tssR > AUGX_NINF;
    case AUGX_K_UTR3SINGLE: case AUGX_K_UTR3TERM: return j == n - 1 || (boep + T.boxlen - 1 < n && ttsPlusOpen(P, boep));
    default: return j + 3 <= n - 1 && P.isRCStop(j + 1);
    }
}

// per-base UTR signal record, end gates of the UTR exon states (OR-ed into B.gate) and the entries of the six site lists.
// Runs after the scans and after k1SiteSignals (it reads SIG_ASSF / SIG_DSSR of the splice sites ending here).
AUGX_HD void k1UtrSignals(const DevTables &T, const BatchView &B, int64_t g, const uint8_t *lcode = nullptr, int lLo = 0, int lHi = 0) {
    const int p = B.chunkPiece[g / CHUNK];
    const int64_t o = B.off[p];
    const int q = (int)(g - o - 1), n = B.len[p];
    double *us = B.usig + g * NUSIG;
    for (int i = 0; i < NUSIG; i++) us[i] = AUGX_NINF;
    if (q < 0 || q >= n || B.cls[p] < 0 || !T.utr) return;
    Piece P = makePieceAt(T, B, p, B.gcPlane[g]); // the class of base q
    P.lcode = lcode; P.lLo = lLo; P.lHi = lHi;
    const int up = T.tss_upwin, te = T.tss_end, dc = T.dpc, bl = T.boxlen, assWhole = T.As + 2 + T.Ae, dssWhole = T.Ds + 2 + T.De;
    {   // forward TSS window beginning at q.  The reference computes the value when a 5' UTR state first asks for it and keeps it
        // (tssProbsPlus, :1788-1790): with the class current THEN, as a rule that of the transcription start -- used here
        int qc = q + up; if (qc > n - 1) qc = n - 1;
        Piece PF = P;
        PF.c = baseClass(B, p, o + 1 + qc);
        us[USIG_TSSF] = tssProbCalc(PF, q, true);
    }
    us[USIG_TSSR] = tssProbCalc(P, q - up - te + 1, false); // the reverse TSS window ENDING at q is asked for at column q (:1098)
    // the window that begins at base 0: entry 0 of tssProbsPlus / tssProbsMinus is neither cleared at a class step (updateToLocalGC
    // clears [from, to) with from = 1, src/utrmodel.cc:779-781) nor re-allocated while the sequences keep one length (:744-747): the
    // value an earlier sequence of this length computed for ITS base 0 answers (the caller knows which, include/augx.h: augx_tss0)
    if (B.tss0) {
        if (q == 0 && B.tss0[2 * p] == B.tss0[2 * p]) us[USIG_TSSF] = B.tss0[2 * p];
        if (q - up - te + 1 == 0 && B.tss0[2 * p + 1] == B.tss0[2 * p + 1]) us[USIG_TSSR] = B.tss0[2 * p + 1];
    }
    us[USIG_TTSP] = ttsPlusCalc(P, q);
    us[USIG_TTSM] = ttsMinusCalc(P, q);
    uint64_t gate = 0;
    if (q >= 1)
        for (int s = 0; s < T.S; s++) {
            if (!T.reachable[s] || !isUtrExonKind(T.kind[s])) continue;
            if (utrGateOpen(P, T.kind[s], q, us[USIG_TSSR])) gate |= 1ull << T.vbit[s];
        }
    B.gate[g] |= gate;
    // site-list entries of the sites at q
    const int64_t lo = listOff(B, p);
    auto pfx = [&](int f, int pos) -> double { // content prefix up to and including base pos (pos < 0: empty)
        if (pos < 0) return 0.0;
        if (pos > n - 1) pos = n - 1;
        return fxD(B.ufx[fidx(o + 1 + pos, f, NUFX)]);
    };
    auto ucn = [&](int f) -> uint32_t { return B.ucnt[fidx(g, f, NUCNT)]; };
    auto ucp = [&](int f) -> uint32_t { return B.ucnt[fidx(g - 1, f, NUCNT)]; };
    if (ucn(UCNT_TF) != ucp(UCNT_TF)) {
        USite &e = B.tfSite[lo + ucn(UCNT_TF) - 1];
        e.pos = q - 1; e.pad = 0; e.b[0] = us[USIG_TSSF] - pfx(UFX_5IF, q + up + te - 1); e.b[1] = e.b[2] = AUGX_NINF;
    }
    if (ucn(UCNT_FS) != ucp(UCNT_FS)) {
        USite &e = B.fsSite[lo + ucn(UCNT_FS) - 1];
        e.pos = q; e.pad = 0; e.b[0] = 0.0 - pfx(UFX_3F, q); e.b[1] = e.b[2] = AUGX_NINF;
    }
    if (ucn(UCNT_TM) != ucp(UCNT_TM)) {
        USite &e = B.tmSite[lo + ucn(UCNT_TM) - 1];
        e.pos = q - dc - 1; e.pad = 0; e.b[0] = us[USIG_TTSM] - pfx(UFX_3R, q + bl - 1); e.b[1] = e.b[2] = AUGX_NINF;
    }
    if (ucn(UCNT_RT) != ucp(UCNT_RT)) {
        USite &e = B.rtSite[lo + ucn(UCNT_RT) - 1];
        e.pos = q; e.pad = 0; e.b[0] = 0.0 - pfx(UFX_5IR, q); e.b[1] = 0.0 - pfx(UFX_5R, q); e.b[2] = AUGX_NINF;
    }
    const uint32_t la1 = B.cnt[fidx(g, CNT_LA, NCNT)], la0 = B.cnt[fidx(g - 1, CNT_LA, NCNT)];
    if (la1 != la0) { // a longass state may end at q: the UTR exon after a UTR intron begins U + As + 2 + Ae - 1 bases before q
        USite &e = B.laSite[lo + la1 - 1];
        const double a = B.sig[g * NSIG + SIG_ASSF]; // aSSProb + the soft-masking bonus of the intronic part (the same bases, :1533-1545)
        e.pos = q - T.U - assWhole; e.pad = 0; e.b[0] = a - pfx(UFX_5F, q); e.b[1] = a - pfx(UFX_3F, q); e.b[2] = AUGX_NINF;
    }
    const uint32_t lr1 = B.cnt[fidx(g, CNT_LR, NCNT)], lr0 = B.cnt[fidx(g - 1, CNT_LR, NCNT)];
    if (lr1 != lr0) { // a reverse longdss state may end at q
        USite &e = B.lrSite[lo + lr1 - 1];
        const double a = B.sig[g * NSIG + SIG_DSSR];
        e.pos = q - dssWhole; e.pad = 0; e.b[0] = a - pfx(UFX_5R, q); e.b[1] = a - pfx(UFX_5IR, q); e.b[2] = a - pfx(UFX_3R, q);
    }
}

// =================================================================================================
// two call-history caches of the reference that UTR states read, replayed on pieces with several GC classes from the aliveness a
// first run left in the dense matrix `mat` ([len][S] of the piece): tssProbsPlus (here, entirely) and the aSSProb memo (assmemo.h
// walks the requests on the host; here: what it reads, and the values it asks to be rebuilt)
// =================================================================================================
AUGX_HD bool memoAnyAlive(const DevTables &T, const double *mat, int S, int s, int eop) { // a request is made for a live predecessor only (src/utrmodel.cc:957-960)
    const int64_t col = eop > 0 ? eop : 0;
    for (int ai = 0; ai < T.n_anc[s]; ai++) if (mat[col * S + T.anc[s][ai]] > AUGX_NINF) return true;
    return false;
}
// forward TSS window of TF site li of piece p: the reference computes tssProb(left) when utr5single or utr5init FIRST ask for it --
// at the first column j whose state passes its end gate, holds left - 1 in its window of predecessor ends and has a live
// predecessor there -- with the class current in that column, and keeps it (tssProbsPlus, src/utrmodel.cc:1788-1790; the entries of
// a class region are forgotten only when the sweep ENTERS the region, :779-780, i.e. before any request for them).
// k1UtrSignals took the class of the transcription start.  true: the value was rebuilt
AUGX_HD bool k1TssReplay(const DevTables &T, const BatchView &B, int p, int li, const double *mat) {
    const int64_t o = B.off[p], lo = listOff(B, p);
    const int n = B.len[p], S = T.S;
    if (B.nPlanes[p] <= 1 || li >= (int)B.ucnt[fidx(o + n, UCNT_TF, NUCNT)]) return false;
    USite &e = B.tfSite[lo + li];
    const int left = e.pos + 1, up = T.tss_upwin, te = T.tss_end;
    if (left == 0 && B.tss0 && B.tss0[2 * p] == B.tss0[2 * p]) return false; // (answered from an earlier sequence: k1UtrSignals)
    int jFirst = -1;
    for (int s = 0; s < S; s++) {
        if (!T.reachable[s] || (T.kind[s] != AUGX_K_UTR5SINGLE && T.kind[s] != AUGX_K_UTR5INIT)) continue;
        if (!memoAnyAlive(T, mat, S, s, left - 1)) continue;
        for (int j = left; j < n && (jFirst < 0 || j < jFirst); j++) {
            int lm, rm;
            utrWindow(T, T.kind[s], j, n, lm, rm);
            if (lm > left - 1) break;
            if (rm < left - 1 || j < 1) continue;
            if ((B.gate[o + 1 + j] >> T.vbit[s]) & 1ull) { jFirst = j; break; }
        }
    }
    if (jFirst < 0) return false;
    int qc = left + up; if (qc > n - 1) qc = n - 1;
    const int plNat = B.gcPlane[o + 1 + qc], plAsk = B.gcPlane[o + 1 + jFirst];
    if (plNat == plAsk) return false;
    Piece PF = makePieceAt(T, B, p, plAsk);
    const double v = tssProbCalc(PF, left, true);
    B.usig[(o + 1 + left) * NUSIG + USIG_TSSF] = v;
    int pe = left + up + te - 1; if (pe > n - 1) pe = n - 1;
    e.b[0] = v - fxD(B.ufx[fidx(o + 1 + pe, UFX_5IF, NUFX)]);
    return true;
}
// what assmemo.h reads of acceptor site li of the LA list (li >= its length: the sites whose AG lies inside the piece while the
// longass state that belongs to them would end after it, q = n ...): q, and per requester whether a predecessor is alive
AUGX_HD int memoAssSite(const DevTables &T, const BatchView &B, int p, int li, const double *mat, const int *reqS, int nReq, uint8_t &alive) {
    const int64_t o = B.off[p], lo = listOff(B, p);
    const int n = B.len[p], S = T.S, nList = (int)B.cnt[fidx(o + n, CNT_LA, NCNT)];
    const int off = T.U + T.As + 2 + T.Ae;
    alive = 0;
    int q;
    if (li < nList) q = B.laPos[lo + li];
    else {
        q = n + (li - nList);
        const uint8_t *code = B.code + o + 1;
        const int pos = q - T.Ae; // isPossibleASS(pos): the AG at pos - 1, pos
        if (!(q <= n - 2 + T.Ae && pos >= 1 && pos <= n - 2 && code[pos - 1] == 0 && code[pos] == 2)) return -1;
    }
    const int eop = q - off;
    if (eop < 0) return -1;
    for (int r = 0; r < nReq; r++)
        if ((li < nList || T.kind[reqS[r]] != AUGX_K_LONGASS) && memoAnyAlive(T, mat, S, reqS[r], eop)) alive |= (uint8_t)(1u << r);
    return q;
}
AUGX_HD uint8_t memoGateBits(const DevTables &T, const BatchView &B, int p, int j, const int *reqS, int nReq) {
    const uint64_t g = B.gate[B.off[p] + 1 + j];
    uint8_t m = 0;
    for (int r = 0; r < nReq; r++) if (T.kind[reqS[r]] != AUGX_K_LONGASS && ((g >> T.vbit[reqS[r]]) & 1ull)) m |= (uint8_t)(1u << r);
    return m;
}
// the values of one acceptor site rebuilt as the replay of the memo asks (dp.h: AssPatch; sw / out: the batch's arrays)
// value of the acceptor site whose longass state would end at q under the class of plane pl, as k1SiteSignals makes SIG_ASSF: aSSProb +
// the soft-masking bonus of the intronic part of the window (a site past the end of the piece: the bases inside it)
AUGX_HD double assSiteValue(const DevTables &T, const BatchView &B, int p, int pl, int q) {
    const int64_t o = B.off[p];
    const int n = B.len[p], begin = q - (T.As + 2 + T.Ae) - T.U + 1;
    Piece P = makePieceAt(T, B, p, pl);
    double v = assProb(P, begin, true);
    if (T.soft && v > AUGX_NINF) {
        const int a = begin < 0 ? 0 : begin;
        int b2 = q - T.Ae; if (b2 > n - 1) b2 = n - 1;
        if (b2 >= a) v = v + (double)(int64_t)((uint64_t)B.cnt[fidx(o + 1 + b2, CNT_SOFT, NCNT)] - (uint64_t)B.cnt[fidx(o + a, CNT_SOFT, NCNT)]) * T.lnSoft;
    }
    return v;
}
AUGX_HD void k1AssPatch(const DevTables &T, const BatchView &B, int p, const AssPatch &A, const AssSwIn *sw, LaSw *out) {
    const int64_t o = B.off[p], lo = listOff(B, p);
    const int q = B.laPos[lo + A.li];
    auto aOf = [&](int pl) -> double { return assSiteValue(T, B, p, pl, q); };
    if (A.longPl >= 0) B.sig[(o + 1 + q) * NSIG + SIG_ASSF] = aOf(A.longPl);
    if (A.basePl < 0) return;
    const double a0 = aOf(A.basePl);
    USite &e = B.laSite[lo + A.li];
    e.b[0] = a0 - fxD(B.ufx[fidx(o + 1 + q, UFX_5F, NUFX)]);
    e.b[1] = a0 - fxD(B.ufx[fidx(o + 1 + q, UFX_3F, NUFX)]);
    e.pad = 0;
    if (A.nSw > 0 && a0 > AUGX_NINF) {
        e.pad = A.swOff + 1;
        for (int k = 0; k < A.nSw; k++) { out[A.swOff + k].key = sw[A.swOff + k].key; out[A.swOff + k].more = (uint32_t)(A.nSw - 1 - k); out[A.swOff + k].cum = aOf(sw[A.swOff + k].pl) - a0; }
    }
}

// =================================================================================================
// candidates of the exon-like UTR states, evaluated where they are needed (trellis, forward, back-trace, sampler)
// =================================================================================================
struct UDesc {          // state s ending at base j
    int8_t kind, list, bsel, fxf, ovl, len, xFirst, nPre;
    int16_t s, pad2;
    int32_t j, nList, nExtra, total;
    int32_t i1;         // one past the newest list entry (piece-local index)
    int32_t xHi;        // predecessor end of the first extra candidate (they run downwards)
    int32_t eom, eobe, cb, cbobe;
    double endP, E;     // ln end signal; endP + content prefix up to the end of the middle part
    // the first nPre listed candidates -- the ones with a middle part of at most one base, the expensive case of utrCandPre --
    // evaluated once by the descriptor kernel (kUtrDesc; 0 where a descriptor is made on the spot: back-trace, sampler)
    int32_t preEop[3], pad3;
    double preTe[3];
};
// read-only view of one piece for the UTR candidates
struct UCtx {
    const DevTables &T;
    const BatchView &B;
    int p, n;
    int64_t o, lo;
    AUGX_HD UCtx(const DevTables &t, const BatchView &b, int pp) : T(t), B(b), p(pp) { n = B.len[p]; o = B.off[p]; lo = listOff(B, p); }
    AUGX_HD uint32_t cntU(int f, int q) const { if (q < 0) return 0; if (q > n - 1) q = n - 1; return B.ucnt[fidx(o + 1 + q, f, NUCNT)]; }
    AUGX_HD uint32_t cntS(int f, int q) const { if (q < 0) return 0; if (q > n - 1) q = n - 1; return B.cnt[fidx(o + 1 + q, f, NCNT)]; }
    AUGX_HD double pfx(int f, int pos) const { if (pos < 0) return 0.0; if (pos > n - 1) pos = n - 1; return fxD(B.ufx[fidx(o + 1 + pos, f, NUFX)]); }
    AUGX_HD int clsAt(int q) const { return baseClass(B, p, o + 1 + q); }
    AUGX_HD const USite *list(int l) const { return (l == UL_TF ? B.tfSite : l == UL_LA ? B.laSite : l == UL_FS ? B.fsSite : l == UL_LR ? B.lrSite : l == UL_TM ? B.tmSite : B.rtSite) + lo; }
};
// descriptor of UTR exon state s ending at j (the gate bit says endPart > 0; the value is computed here).  total == 0: nothing to do
AUGX_HD void utrDescribe(const UCtx &X, int s, int j, UDesc &D) {
    const DevTables &T = X.T;
    const BatchView &B = X.B;
    const int kind = T.kind[s], n = X.n;
    const UGeom g = utrGeom(T, kind);
    D.kind = (int8_t)kind; D.list = g.list; D.bsel = g.bsel; D.fxf = g.fxf; D.ovl = g.ovl; D.len = g.len; D.cb = g.cb; D.cbobe = g.cbobe;
    D.s = (int16_t)s; D.j = j; D.nList = D.nExtra = D.total = 0; D.i1 = 0; D.xHi = 0; D.xFirst = 0; D.nPre = 0; D.pad2 = 0; D.pad3 = 0;
    for (int i = 0; i < 3; i++) { D.preEop[i] = 0; D.preTe[i] = AUGX_NINF; }
    int boep, eobe, lm, rm;
    utrEndPos(T, kind, j, n, boep, eobe);
    utrWindow(T, kind, j, n, lm, rm);
    D.eom = boep - 1; D.eobe = eobe;
    const int assWhole = T.As + 2 + T.Ae;
    const int64_t gj = X.o + 1 + j;
    double endP = 0.0;
    switch (kind) { // endPartEmiProb :1072-1161 (with the soft-masking bonus of the intronic part of the state, :1144-1157)
    case AUGX_K_UTR5SINGLE: case AUGX_K_UTR5TERM: break; // (the gate was the start codon)
    case AUGX_K_UTR5INIT: case AUGX_K_UTR5INTERNAL: case AUGX_K_UTR3INIT: case AUGX_K_UTR3INTERNAL: {
        Piece P = makePiece(T, B, X.p);
        endP = dssProb(P, boep, true);
        break;
    }
    case AUGX_K_RUTR5INTERNAL: case AUGX_K_RUTR5TERM: case AUGX_K_RUTR3INTERNAL: case AUGX_K_RUTR3TERM:
        if (boep >= 1) endP = B.sig[gj * NSIG + SIG_ASSR]; // (= aSSProb of the window + the bonus of the same intronic bases)
        else { // the window begins at base 0: the rlongass state cannot end here, the UTR exon can
            Piece P = makePieceAt(T, B, X.p, B.gcPlane[gj]);
            endP = assProb(P, boep, false);
        }
        break;
    case AUGX_K_RUTR5SINGLE: case AUGX_K_RUTR5INIT: endP = B.usig[gj * NUSIG + USIG_TSSR]; break;
    case AUGX_K_UTR3SINGLE: case AUGX_K_UTR3TERM: endP = j == n - 1 ? 0.0 : B.usig[(X.o + 1 + boep) * NUSIG + USIG_TTSP]; break;
    default: break; // RUTR3SINGLE, RUTR3INIT: the gate was the reverse stop codon
    }
    if (T.soft && eobe < j && endP > AUGX_NINF &&
        (kind == AUGX_K_UTR5INIT || kind == AUGX_K_UTR5INTERNAL || kind == AUGX_K_UTR3INIT || kind == AUGX_K_UTR3INTERNAL)) {
        const int a = eobe + 1 < 0 ? 0 : eobe + 1;
        endP = endP + (double)(int64_t)((uint64_t)X.cntS(CNT_SOFT, j) - (uint64_t)X.cntS(CNT_SOFT, a - 1)) * T.lnSoft;
    }
    if (T.soft && boep < 1 && endP > AUGX_NINF && eobe < j &&
        (kind == AUGX_K_RUTR5INTERNAL || kind == AUGX_K_RUTR5TERM || kind == AUGX_K_RUTR3INTERNAL || kind == AUGX_K_RUTR3TERM)) {
        const int a = eobe + 1 < 0 ? 0 : eobe + 1;
        endP = endP + (double)(int64_t)((uint64_t)X.cntS(CNT_SOFT, j) - (uint64_t)X.cntS(CNT_SOFT, a - 1)) * T.lnSoft;
    }
    D.endP = endP;
    if (!(endP > AUGX_NINF) || rm < lm) return;
    D.E = endP + X.pfx(g.fxf, D.eom);
    // listed candidates: the sites of the kind's list whose predecessor end lies in [lm, rm]
    int sLo, sHi, fld = 0; // site positions (in the list's own coordinate) of the window
    bool ownCnt = true;
    switch (g.list) {
    case UL_TF: sLo = (lm + 1 < 0 ? 0 : lm + 1); sHi = rm + 1; fld = UCNT_TF; break;                   // by window begin = eop + 1
    case UL_LA: sLo = lm + T.U + assWhole; sHi = rm + T.U + assWhole; fld = CNT_LA; ownCnt = false; break; // by longass end
    case UL_FS: sLo = lm; sHi = rm; fld = UCNT_FS; break;
    case UL_LR: sLo = lm + T.Ds + 2 + T.De; sHi = rm + T.Ds + 2 + T.De; fld = CNT_LR; ownCnt = false; break;
    case UL_TM: sLo = (lm < 0 ? 0 : lm) + 1 + T.dpc; sHi = rm + 1 + T.dpc; fld = UCNT_TM; break;       // by box begin = eop + 1 + dc
    default: sLo = lm; sHi = rm; fld = UCNT_RT;
    }
    if (sHi >= sLo && sHi >= 0) {
        const int64_t i0 = ownCnt ? (int64_t)X.cntU(fld, sLo - 1) : (int64_t)X.cntS(fld, sLo - 1);
        D.i1 = (int32_t)(ownCnt ? X.cntU(fld, sHi) : X.cntS(fld, sHi));
        D.nList = (int)((int64_t)D.i1 - i0);
        if (D.nList < 0) D.nList = 0;
    }
    // candidates that are not sites of a list: truncated begins before the piece (TF, TM) and the start from column 0 (FS, RT)
    if (g.list == UL_TF) { const int hi = rm < -2 ? rm : -2; if (hi >= lm) { D.nExtra = hi - lm + 1; D.xHi = hi; } }
    else if (g.list == UL_TM) { const int hi = rm < -1 ? rm : -1; if (hi >= lm) { D.nExtra = hi - lm + 1; D.xHi = hi; } }
    else if (g.list == UL_FS || g.list == UL_RT) { if (lm <= 0 && rm >= 0) { D.nExtra = 1; D.xHi = 0; } }
    else if (g.list == UL_LA) {
        // an acceptor site whose AG lies inside the piece while the longass state that would end Ae bases after it does not: not on
        // the list, but a possible begin of the UTR exon (aSSProb then scores the cut-off pattern as unknown, src/intronmodel.cc:1178-1180).
        // These candidates have the largest predecessor ends of the window: they come first
        const int qLo = sLo > n ? sLo : n, qHi = sHi < n - 2 + T.Ae ? sHi : n - 2 + T.Ae;
        if (qHi >= qLo) { D.nExtra = qHi - qLo + 1; D.xHi = qHi - (T.U + assWhole); D.xFirst = 1; }
    }
    D.total = D.nList + D.nExtra;
}
AUGX_HD double utrLenAt(const DevTables &T, int sel, int len, bool tail3) {
    if (len < 0) return AUGX_NINF;
    if (tail3) return len <= T.uM3S ? AUGX_GTAB(T.tail3s)[len] : AUGX_NINF;
    switch (sel) {
    case 0: return len <= T.uML ? AUGX_GTAB(T.len5s)[len] : AUGX_NINF;
    case 1: return len <= T.uML ? AUGX_GTAB(T.len5i)[len] : AUGX_NINF;
    case 2: return len <= T.uML ? AUGX_GTAB(T.len5n)[len] : AUGX_NINF;
    case 3: return len <= T.uML ? AUGX_GTAB(T.len5t)[len] : AUGX_NINF;
    case 4: return len <= T.uM3S ? AUGX_GTAB(T.len3s)[len] : AUGX_NINF;
    case 5: return len <= T.uML ? AUGX_GTAB(T.len3i)[len] : AUGX_NINF;
    case 6: return len <= T.uML ? AUGX_GTAB(T.len3n)[len] : AUGX_NINF;
    default: return len <= T.uM3T ? AUGX_GTAB(T.len3t)[len] : AUGX_NINF;
    }
}
// single-base middle part: SegProbs::getSeqProb with from == to reads the table of the class CURRENT at the time, i.e. of the end
// base j of the state (src/statemodel.cc:437-449)
AUGX_HD double utrEmi1(const UCtx &X, int fxf, int cj, int pos) {
    const DevTables &T = X.T;
    const double *tab = AUGX_GTAB(fxf <= UFX_5IR ? T.utr5init_emi : fxf <= UFX_5R ? T.utr5_emi : T.utr3_emi) + (int64_t)cj * T.uNP;
    const bool fwd = (fxf & 1) == 0;
    const uint8_t *code = X.B.code + X.o + 1;
    auto bb = [&](int q) -> int { return (q >= 0 && q < X.n) ? code[q] : 4; };
    int r = 0;
    if (fwd) {
        if (pos < T.uk) return T.ln_quarter;
        for (int i = 0; i <= T.uk; i++) { const int c = bb(pos - T.uk + i); if (c > 3) return T.ln_quarter; r = (r << 2) | c; }
    } else
        for (int i = 0; i <= T.uk; i++) { const int c = bb(pos + i); if (c > 3) return T.ln_quarter; r |= (3 - c) << (2 * i); }
    return tab[r];
}
// candidate idx (0 = the largest predecessor end) of the state described by D: te = ln emission of the state from eop + 1 to j
// (without the transition term), eop = end of the predecessor.  false: infeasible
// (candidate idx is entry D.i1 - 1 - li of the kind's site list, or -- xi >= 0 -- the xi-th candidate that is not on a list)
AUGX_HD void utrCandIndex(const UDesc &D, int idx, int &li, int &xi) {
    li = idx; xi = -1;
    if (D.xFirst) { if (idx < D.nExtra) xi = idx; else li = idx - D.nExtra; }
    else if (idx >= D.nList) xi = idx - D.nList;
}
// everything of the candidate but its length term: sig = begin + middle + end part, len / tail3 = argument of the length distribution
// (sitePos, siteB: predecessor end and (begin signal - content prefix) of the candidate's site record, for a listed candidate)
// Returns 0: infeasible, 1: done, 2 (deferRare only): one of the rare, expensive cases -- a candidate that is not on a site list, or
// a middle part of at most one base -- which the caller evaluates apart from the common ones.
AUGX_HD int utrCandPre(const UCtx &X, const UDesc &D, int xi, int sitePos, double siteB, double &sig, int &len, bool &tail3, int &eop, bool deferRare = false) {
    const DevTables &T = X.T;
    const int n = X.n;
    double bmix, braw = 0.0; // (begin signal) - (content prefix before the middle part); the begin signal alone
    bool haveRaw = false;
    if (deferRare && xi >= 0) return 2;
    if (xi < 0) {
        eop = sitePos;
        bmix = siteB;
    } else {
        eop = D.xHi - xi;
        const int begin = eop + 1, bom = begin + D.cb;
        // part of the begin signal lies before the piece (:1190-1194,1206-1210,1304-1311,1385-1389); a start from column 0
        // of a state without begin signal; an acceptor site at the very end of the piece
        if (D.list == UL_LA) {
            Piece P = makePieceAt(T, X.B, X.p, X.B.gcPlane[X.o + n]); // (the class of the last base)
            braw = assProb(P, begin, true);
            if (T.soft && braw > AUGX_NINF) {
                int hi = begin + D.cbobe - 1; if (hi > n - 1) hi = n - 1;
                braw = braw + (double)(int64_t)((uint64_t)X.cntS(CNT_SOFT, hi) - (uint64_t)X.cntS(CNT_SOFT, begin - 1)) * T.lnSoft;
            }
        } else if (D.list == UL_TF) braw = (bom - 1) * T.ln_quarter;
        else if (D.list == UL_TM) braw = (D.kind == AUGX_K_RUTR3TERM || bom > 0) ? (bom - 1) * T.ln_quarter : 0.0;
        else braw = 0.0;
        haveRaw = true;
        bmix = braw - X.pfx(D.fxf, bom - 1);
    }
    if (!(bmix > AUGX_NINF)) return 0;
    const int begin = eop + 1, bom = begin + D.cb, mlen = D.eom - bom + 1;
    if (deferRare && mlen <= 1) return 2;
    if (mlen > 1) sig = bmix + D.E; // begin part + middle part + end part
    else {
        if (!haveRaw) braw = bmix + X.pfx(D.fxf, bom - 1);
        double mp;
        if (mlen == 1) mp = utrEmi1(X, D.fxf, X.clsAt(D.j), D.eom);
        else if (mlen == 0) mp = 0.0;
        else mp = D.ovl == 1 ? -mlen * T.ln2 : D.ovl == 2 ? -mlen * T.ln4 : 0.0;
        sig = (braw + mp) + D.endP;
    }
    const int bobe = begin + D.cbobe;
    len = D.eobe - bobe + 1;
    tail3 = false;
    if ((D.kind == AUGX_K_UTR3SINGLE || D.kind == AUGX_K_UTR3TERM) && D.eobe == n - 1) tail3 = true; // right-truncated 3' UTR (:1290-1295,1369-1372)
    if (D.kind == AUGX_K_RUTR3SINGLE && begin <= 0) tail3 = true;                                     // left-truncated (:1312)
    return 1;
}
// (begin signal - content prefix) of site record e for the state described by D.  An acceptor site whose value changes during the
// sweep (assmemo.h; e.pad > 0, rare: near a GC-class step) carries a table of changes by (end base, state) of the asking state
AUGX_HD double utrSiteB(const BatchView &B, const UDesc &D, const USite &e) {
    double bv = D.bsel == 0 ? e.b[0] : D.bsel == 1 ? e.b[1] : e.b[2];
    if (D.list == UL_LA && e.pad > 0) {
        const uint32_t key = ((uint32_t)D.j << 7) | (uint32_t)D.s;
        const LaSw *w = B.laSw + (e.pad - 1);
        double cum = 0.0;
        for (;;) {
            if (key < w->key) break;
            cum = w->cum;
            if (!w->more) break;
            w++;
        }
        bv = bv + cum;
    }
    return bv;
}
AUGX_HD bool utrCandFrom(const UCtx &X, const UDesc &D, int xi, const USite &e, double &te, int &eop) {
    double sig; int len; bool tail3;
    if (!utrCandPre(X, D, xi, e.pos, utrSiteB(X.B, D, e), sig, len, tail3, eop)) return false;
    const double lp = utrLenAt(X.T, D.len, len, tail3);
    if (!(lp > AUGX_NINF)) return false;
    te = sig + lp;
    return te > AUGX_NINF;
}
AUGX_HD bool utrCand(const UCtx &X, const UDesc &D, int idx, double &te, int &eop) {
    int li, xi;
    utrCandIndex(D, idx, li, xi);
    USite e;
    e.pos = 0; e.pad = 0; e.b[0] = e.b[1] = e.b[2] = AUGX_NINF;
    if (xi < 0) e = X.list(D.list)[D.i1 - 1 - li];
    return utrCandFrom(X, D, xi, e, te, eop);
}

// =================================================================================================
// the state graph as the dense kernels see it
// =================================================================================================
constexpr int DCH = 16;   // chain-state slots (intergenic, geometric introns, UTR introns)
constexpr int DFIX = 24;  // fixed-lag states
constexpr int DUV = 16;   // exon-like UTR states
AUGX_HD bool isChainKind(int k) { return k == AUGX_K_IGENIC || k == AUGX_K_GEOMETRIC || k == AUGX_K_RGEOMETRIC || isUtrIntronKind(k); }
AUGX_HD bool isEarlyChainKind(int k) { return k == AUGX_K_GEOMETRIC || k == AUGX_K_RGEOMETRIC; }
AUGX_HD bool isFixedKind(int k) { return k == AUGX_K_LONGDSS || k == AUGX_K_RLONGDSS || k == AUGX_K_LONGASS || k == AUGX_K_RLONGASS || k == AUGX_K_EQUALD || k == AUGX_K_REQUALD; }
AUGX_HD bool isItemKind(int k) { return (k >= AUGX_K_SINGLE && k <= AUGX_K_RTERMINAL) || k == AUGX_K_LESSD || k == AUGX_K_RLESSD; }
AUGX_HD double initLn(const DevTables &T, int initKind, int s) { return initKind == 0 ? T.ln_init[s] : (s == T.synch ? 0.0 : AUGX_NINF); }

// a value that is the same in every lane of the wavefront, moved to a scalar register: what is computed from it is scalar
// arithmetic, a branch on it a scalar branch
#ifdef AUGX_EMU
inline int uni(int v) { return v; }
inline UDesc uniDesc(const UDesc &d) { return d; }
#else
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ UDesc uniDesc(const UDesc &d) {
    union { UDesc d; int w[sizeof(UDesc) / 4]; } a, r;
    a.d = d;
#pragma unroll
    for (int i = 0; i < (int)(sizeof(UDesc) / 4); i++) r.w[i] = __builtin_amdgcn_readfirstlane(a.w[i]);
    return r.d;
}
#endif
#ifdef AUGX_EMU
inline USite ldUSite(const USite *p) { return *p; }
#else
__device__ __forceinline__ USite ldUSite(const USite *p) { // two 16-byte global loads
    typedef int v4i __attribute__((ext_vector_type(4)));
    const v4i a = ((const AUGX_GLOBAL v4i *)p)[0], c = ((const AUGX_GLOBAL v4i *)p)[1];
    USite e;
    e.pos = a.x; e.pad = a.y; e.b[0] = __hiloint2double(a.w, a.z); e.b[1] = __hiloint2double(c.y, c.x); e.b[2] = __hiloint2double(c.w, c.z);
    return e;
}
#endif
constexpr int UDW = 14;    // 64-bit words of a descriptor
constexpr int UDCAP = 64;  // descriptors of a block staged in LDS (a block with more reads the rest from HBM)
static_assert(sizeof(UDesc) == UDW * 8, "UDesc is copied word by word");

// ---- descriptors of the open (end base, UTR exon state) pairs, once per decode: they depend on the sequence only, not on ln V.
// One workgroup per 32 bases, thread = (base, state); the descriptors with candidates are packed, block by block, into B.ud
// (space handed out by one atomic add per workgroup, like the candidate records of kCand: the order of the blocks in the buffer is
// whatever it happens to be, the trellis takes maxima and sums integers).
struct UDescLds { int has[NT]; int gpre[WAVE + 1]; unsigned long long base; };
template <int BLK>
AUGX_KFN void utrDescGroup(const DevTables &T, const BatchView &B, UDescLds &L, int64_t wg) {
    constexpr int NBASE = NT / 16;
    static_assert(DUV == 16 && NBASE % BLK == 0 && (BLK * 16) % 8 == 0, "thread = (base, slot)");
    const int64_t g0 = wg * NBASE; // slot off + j of the group's first base
    const int p = B.chunkPiece[g0 / CHUNK];
    const int64_t o = B.off[p];
    const int n = B.len[p], j0 = (int)(g0 - o);
    if (!T.utr || j0 >= n || B.cls[p] < 0) return; // (blocks past the end of a piece are never read)
    UCtx X(T, B, p);
    TV(UDesc, dd);
    FOR_THREADS(t) {
        const int dj = t / 16, slot = t % 16, j = j0 + dj;
        UDesc &D = TX(dd);
        D.total = 0;
        const int s2 = slot < T.nUv ? T.uvS[slot] : -1;
        if (s2 >= 0 && j >= 1 && j < n && ((B.gate[o + 1 + j] >> T.vbit[s2]) & 1ull)) {
            utrDescribe(X, s2, j, D);
            for (int li = 0; li < 3 && li < D.nList; li++) { // the leading listed candidates while they are of the expensive kind
                const int idx = D.xFirst ? D.nExtra + li : li;
                const USite e = X.list(D.list)[D.i1 - 1 - li];
                double sig; int len, eop; bool tail3;
                if (utrCandPre(X, D, -1, e.pos, utrSiteB(B, D, e), sig, len, tail3, eop, true) != 2) break;
                double te = AUGX_NINF;
                if (!utrCand(X, D, idx, te, eop)) te = AUGX_NINF;
                D.preEop[li] = e.pos; D.preTe[li] = te; D.nPre = (int8_t)(li + 1);
            }
        }
        L.has[t] = D.total > 0;
    }
    BLOCK_SYNC();
    TV(int, g8);
    FOR_THREADS(t) {
        int c = 0;
        if (t < WAVE) for (int k = 0; k < 8; k++) c += L.has[t * 8 + k];
        TX(g8) = c;
    }
    FOR_WAVES(w) { if (w == 0) waveInclScan(g8, w); }
    FOR_THREADS(t) { if (t < WAVE) L.gpre[t + 1] = TX(g8); if (t == 0) L.gpre[0] = 0; }
    BLOCK_SYNC();
    FOR_THREADS(t) {
        if (t == 0) {
            const unsigned long long total = (unsigned long long)L.gpre[WAVE];
#ifdef AUGX_EMU
            L.base = B.candAlloc->descs; B.candAlloc->descs += total;
#else
            L.base = atomicAdd(&B.candAlloc->descs, total);
#endif
        }
    }
    BLOCK_SYNC();
    const uint64_t base = L.base;
    const bool fits = base + (uint64_t)L.gpre[WAVE] <= (uint64_t)B.udCap; // (else the host re-runs the kernel with a buffer of the size the counter reports)
    FOR_THREADS(t) {
        int rank = L.gpre[t / 8];
        for (int k = 0; k < t % 8; k++) rank += L.has[(t / 8) * 8 + k];
        if (fits && L.has[t]) B.ud[base + (uint64_t)rank] = TX(dd);
        if (t % (BLK * 16) == 0) {
            const int64_t gb = o / BLK + (j0 + t / 16) / BLK;
            B.udOff[gb] = base + (uint64_t)L.gpre[t / 8];
            B.udCnt[gb] = fits ? (uint32_t)(L.gpre[(t + BLK * 16) / 8] - L.gpre[t / 8]) : 0u;
        }
    }
}

struct DenseLds {
    double ring[WAVE][SPX];          // the newest 64 columns, [j & 63][state]
    double cmax[2][8][SPX];             // variable-length cells of the block ([parity]): largest candidate ...
    unsigned long long csum[2][8][SPX]; // ... (forward) sum of exp(candidate - largest), fixed point
    double tr[SPX][AUGX_MAX_ANC];    // ln t(ancestor ai -> s) of the piece's first class
    uint8_t anc[SPX][AUGX_MAX_ANC], nanc[SPX];
    uint8_t cellKind[SPX];           // 1: candidates from the records of kCand, 2: reverse terminal exon, 3: UTR exon, 0: none
    double sg[2][8][NSIG];           // signal records of the block ([parity]; the next block's are staged meanwhile)
    uint64_t bOff[2];
    uint32_t bCnt[2][2];
    uint64_t uOff[2];                // descriptors of the block's UTR exon cells: first, count
    uint32_t uCnt[2];
    int chS[DCH], chNa[DCH], chAnc[DCH][AUGX_MAX_ANC], chNd[DCH], chDead[DCH][AUGX_MAX_ANC], chDeadAi[DCH][AUGX_MAX_ANC], chSgi[DCH];
    uint8_t chLive[DCH][AUGX_MAX_ANC], chEarly[DCH];
    double oth[DCH][8];              // [slot][base of the block]: what reaches the chain state from the other states
    uint8_t othAi[DCH][8];
    uint64_t ud[2][UDCAP * UDW];     // the descriptors themselves ([parity])
    // the tables a UTR candidate reads, by number (a pointer taken from the model or batch structure is a scalar load from HBM and a
    // wait each time): the eight length distributions + the tail distribution of the truncated 3' UTR, their last entries; the six site lists
    const double *lenTab[9];
    int lenMax[9];
    const USite *siteTab[6];
};

// value of state a at base q for the block that begins at jb: from the ring while no base of the block has taken its column
template <bool FWD> AUGX_KFN double denseAt(const DenseLds &L, const double *M, int S, int q, int a, int jb, int BLK) {
    if (q < 0) q = 0;
    if (jb + BLK - 1 - q < WAVE) return ldsLoadD(&L.ring[q & 63][a]);
    return ldCoherent(&M[(int64_t)q * S + a]);
}

#if defined(AUGX_EMU) || !defined(AUGX_PROFILE)
#define DPROF(k) do {} while (0)
#else // (developer build -DAUGX_PROFILE, AUGX_PROF=1: cycles of the first lane of wavefront 1 per stage; it passes every barrier)
#define DPROF(k) do { if (B.prof && threadIdx.x == WAVE) { const uint64_t now_ = clock64(); dpAcc[k] += now_ - dpLast; dpLast = now_; } } while (0)
#endif
// One workgroup walks piece p block by block.  MODE 0: Viterbi (ln V into B.cells, back pointers of the chain / fixed-lag
// states into B.bpD, score and final state of the piece); MODE 1: forward algorithm (ln F into B.fwd, ln P(sequence)).
template <int BLK, int MODE, bool TIES = false> // TIES (MODE 0): the chain runs flag near ties (dp.h: AUGX_NEAR_TIE) -- a build of its own
AUGX_KFN void densePiece(const DevTables &T, const BatchView &B, DenseLds &L, int p) {
    constexpr bool FWD = MODE == 1;
    const int n = B.len[p], S = T.S, c0 = B.cls[p];
    const int64_t o = B.off[p];
    double *M = (FWD ? B.fwd : B.cells) + (o + 1) * S;
    uint8_t *BP = B.bpD ? B.bpD + (o + 1) * S : nullptr;
    const double *gSig = B.sig + (o + 1) * NSIG;
    const uint64_t *gUdOff = B.udOff;
    const uint32_t *gUdCnt = B.udCnt;
    const uint64_t *gUd = (const uint64_t *)B.ud;
    const Item *gItems = B.items;
    const uint64_t *gBlkOff = B.blkOff;
    const uint32_t *gBlkCnt = B.blkCnt, *gBlkSplit = B.blkSplit;
    const uint8_t *gPlane = B.gcPlane + o + 1;
    const int32_t *gPlaneCls = B.planeCls + p * MAXPL;
    const int initKind = B.initKind[p], termKind = B.termKind[p], synch = T.synch;
    if (c0 < 0) { FOR_THREADS(t) { if (t == 0) { if (FWD) B.lnFwd[p] = AUGX_NINF; else { B.lnv[p] = AUGX_NINF; B.status[p] = AUGX_E_HIP; B.finalState[p] = -1; } } } return; }
    const bool multi = B.nPlanes[p] > 1;
    const int dssWhole = T.Ds + 2 + T.De, assLag = T.As + 2 + T.Ae + T.U, dL = T.dStateLen;
    auto clsAt = [&](int j) __attribute__((always_inline)) { return multi ? (int)gp(gPlaneCls)[gp(gPlane)[j]] : c0; };
    const double *gTrans = T.ln_trans;
    // ln t(a -> s2), a = ancestor ai of s2, with the class of the end base
    auto trn = [&](int cc, int s2, int ai) __attribute__((always_inline)) -> double {
        if (multi) return gp(gTrans)[((int64_t)cc * S + (*lp(&L.anc[s2][ai]))) * S + s2];
        return ldsLoadD(&L.tr[s2][ai]);
    };
    UCtx UX(T, B, p);
    bool anyNuc = false;
    // ---- tables of the state graph in LDS; the matrix starts empty
    FOR_THREADS(t) {
        for (int i = t; i < WAVE * SPX; i += NT) (*lp(&L.ring[i / SPX][i % SPX])) = AUGX_NINF;
        for (int i = t; i < 8 * SPX; i += NT) { (*lp(&L.cmax[0][i / SPX][i % SPX])) = AUGX_NINF; (*lp(&L.csum[0][i / SPX][i % SPX])) = 0ull; }
        if (t < SPX) {
            const int k = t < S && T.reachable[t] ? T.kind[t] : -1;
            (*lp(&L.cellKind[t])) = k < 0 ? 0 : k == AUGX_K_RTERMINAL ? 2 : isItemKind(k) ? 1 : isUtrExonKind(k) ? 3 : 0;
            (*lp(&L.nanc[t])) = (uint8_t)(t < S ? T.n_anc[t] : 0);
            for (int ai = 0; ai < AUGX_MAX_ANC; ai++) {
                const int a = (t < S && ai < T.n_anc[t]) ? T.anc[t][ai] : 0;
                (*lp(&L.anc[t][ai])) = (uint8_t)a;
                (*lp(&L.tr[t][ai])) = (t < S && ai < T.n_anc[t]) ? lnT(T, c0, a, t) : AUGX_NINF;
            }
        }
        for (int64_t i = t; i < (int64_t)n * S; i += NT) gp(M)[i] = AUGX_NINF; // (absent cells stay -inf)
        if (!FWD && BP) for (int64_t i = t; i < (int64_t)n * S; i += NT) gp(BP)[i] = 0xFF;
    }
    for (int q = 0; q < n && !anyNuc; q++) anyNuc = B.code[o + 1 + q] < 4; // (uniform; the pieces that are all N are rare and short-circuited below)
    BLOCK_GLOBAL_SYNC();
    FOR_THREADS(t) { // column 0 = initial probabilities (reference NAMGene::setStatesInitialProbs, src/namgene.cc:144-150)
        if (t < S) {
            const double v = initLn(T, initKind, t);
            (*lp(&L.ring[0][t])) = v;
            gp(M)[t] = v;
        }
    }
    BLOCK_GLOBAL_SYNC();
    // ---- the states by role (uniform)
    int fixS[DFIX], nFix = 0, chS[DCH], nCh = 0, nEarly = 0;
    const int nUv = T.nUv;
    for (int s2 = 0; s2 < S; s2++) {
        if (!T.reachable[s2]) continue;
        const int k = T.kind[s2];
        if (isFixedKind(k)) { if (nFix < DFIX) fixS[nFix++] = s2; }
        else if (isEarlyChainKind(k)) { if (nCh < DCH) chS[nCh++] = s2; }
    }
    nEarly = nCh;
    for (int s2 = 0; s2 < S; s2++) {
        if (!T.reachable[s2]) continue;
        const int k = T.kind[s2];
        if (isChainKind(k) && !isEarlyChainKind(k)) { if (nCh < DCH) chS[nCh++] = s2; }
    }
    if (!anyNuc) { // all N: everything is intergenic (reference src/namgene.cc:205-226); one thread, column after column
        FOR_THREADS(t) {
            if (t == 0) {
                double v = initLn(T, initKind, synch);
                for (int j = 1; j < n; j++) { v = v - T.ln4; gp(M)[(int64_t)j * S + synch] = v; if (!FWD && BP) gp(BP)[(int64_t)j * S + synch] = 0xFE; }
                const double tl = termKind == 0 ? T.ln_term[synch] : 0.0;
                if (FWD) B.lnFwd[p] = v + tl;
                else { B.lnv[p] = v + tl; B.finalState[p] = (v + tl) > AUGX_NINF ? synch : -1; B.status[p] = (v + tl) > AUGX_NINF ? 0 : AUGX_E_NOPATH; }
            }
        }
        return;
    }
    const int nBlocks = (n + BLK - 1) / BLK;
    const int64_t gb0 = o / BLK;
    // per-thread constants of the fixed-lag step (thread = (state, base of the block))
    // stage 1: the threads below A1T make the fixed-lag states and stage the next block; the wavefronts from UW0 on take the UTR units
    // (blocks of up to 4 bases: one wavefront for the former, two (state, base) pairs per thread, seven for the latter)
    constexpr int UW0 = BLK == 8 ? 3 : 1, A1T = UW0 * WAVE, FR = (DFIX * BLK + A1T - 1) / A1T;
    static_assert(BLK * NSIG <= A1T, "roles of stage 1");
    constexpr int UH = 2; // candidates a lane of a UTR unit has in flight (three spill registers)
    TV2(int, fS2, FR); TV2(int, fLag, FR); TV2(int, fSg, FR);
    // per-thread constants of the chain runs (thread = slot): the state, its own ancestor index, whether the only chain state of its
    // stage among its ancestors is the state itself (then its run over the block is a recurrence in registers)
    TV(int, cS2); TV(int, cSelf); TV(int, cFast); TV(int, cSgi);
    FOR_THREADS(t) {
        TX(cS2) = -1; TX(cSelf) = -1; TX(cFast) = 0; TX(cSgi) = SIG_EIN;
        for (int r = 0; r < FR; r++) {
            const int f = t + r * A1T;
            fS2[r][TI] = -1; fLag[r][TI] = 1; fSg[r][TI] = 0;
            if (t < A1T && f < nFix * BLK) {
                const int s2 = fixS[f / BLK], k = T.kind[s2];
                fS2[r][TI] = s2;
                fLag[r][TI] = (k == AUGX_K_LONGDSS || k == AUGX_K_RLONGDSS) ? dssWhole : (k == AUGX_K_LONGASS || k == AUGX_K_RLONGASS) ? assLag : dL;
                fSg[r][TI] = k == AUGX_K_LONGDSS ? SIG_DSSF : k == AUGX_K_RLONGDSS ? SIG_DSSR : k == AUGX_K_LONGASS ? SIG_ASSF : k == AUGX_K_RLONGASS ? SIG_ASSR : SIG_EQD;
            }
        }
        if (t == NT - 1 && T.utr) {
            const double *lt[9] = {T.len5s, T.len5i, T.len5n, T.len5t, T.len3s, T.len3i, T.len3n, T.len3t, T.tail3s};
            const int lm[9] = {T.uML, T.uML, T.uML, T.uML, T.uM3S, T.uML, T.uML, T.uM3T, T.uM3S};
            for (int i = 0; i < 9; i++) { (*lp(&L.lenTab[i])) = lt[i]; (*lp(&L.lenMax[i])) = lm[i]; }
            for (int i = 0; i < 6; i++) (*lp(&L.siteTab[i])) = UX.list(i);
        }
        if (t < DCH) {
            const int s2 = t < nCh ? chS[t] : -1;
            (*lp(&L.chS[t])) = s2;
            const int na = s2 >= 0 ? T.n_anc[s2] : 0;
            (*lp(&L.chNa[t])) = na;
            (*lp(&L.chEarly[t])) = t < nEarly;
            (*lp(&L.chSgi[t])) = (s2 >= 0 && T.kind[s2] == AUGX_K_IGENIC) ? SIG_EIG : (s2 >= 0 && T.uk != T.k && isUtrIntronKind(T.kind[s2])) ? SIG_EUIN : SIG_EIN;
            int nd = 0, nLive = 0, selfAi = -1;
            bool onlySelf = true;
            for (int ai = 0; ai < AUGX_MAX_ANC; ai++) {
                const int a = ai < na ? T.anc[s2][ai] : 0;
                // an ancestor that is a chain state of the same stage is made in the same run (as a rule only the state itself)
                const bool live = ai < na && isChainKind(T.kind[a]) && (isEarlyChainKind(T.kind[a]) == (t < nEarly));
                (*lp(&L.chAnc[t][ai])) = a;
                (*lp(&L.chLive[t][ai])) = live;
                if (live) { nLive++; if (a == s2) selfAi = ai; else onlySelf = false; }
                if (ai < na && !live) { (*lp(&L.chDead[t][nd])) = a; (*lp(&L.chDeadAi[t][nd])) = ai; nd++; }
            }
            for (int k2 = nd; k2 < AUGX_MAX_ANC; k2++) { (*lp(&L.chDead[t][k2])) = 0; (*lp(&L.chDeadAi[t][k2])) = 0; }
            (*lp(&L.chNd[t])) = nd;
            TX(cS2) = s2; TX(cSelf) = selfAi; TX(cFast) = s2 >= 0 && onlySelf && nLive <= 1;
            TX(cSgi) = (s2 >= 0 && T.kind[s2] == AUGX_K_IGENIC) ? SIG_EIG : (s2 >= 0 && T.uk != T.k && isUtrIntronKind(T.kind[s2])) ? SIG_EUIN : SIG_EIN;
        }
    }
    constexpr int NTW = NT - WAVE;
    FOR_THREADS(t) { // block 0: offsets, signal records, gates
        if (t == NT - 1) { (*lp(&L.bOff[0])) = gp(gBlkOff)[gb0 * 2 + 1]; (*lp(&L.bCnt[0][0])) = gp(gBlkCnt)[gb0 * 2 + 1]; (*lp(&L.bCnt[0][1])) = gp(gBlkSplit)[gb0 * 3 + 2]; }
        if (t >= NT - BLK * NSIG) { const int i = t - (NT - BLK * NSIG); (*lp(&L.sg[0][i / NSIG][i % NSIG])) = i / NSIG < n ? gp(gSig)[(int64_t)(i / NSIG) * NSIG + i % NSIG] : AUGX_NINF; }
        if (nUv > 0) { // the descriptors of block 0
            const uint64_t uo = gp(gUdOff)[gb0];
            const uint32_t uc = gp(gUdCnt)[gb0];
            if (t == 0) { (*lp(&L.uOff[0])) = uo; (*lp(&L.uCnt[0])) = uc; }
            const uint32_t words = (uc < (uint32_t)UDCAP ? uc : (uint32_t)UDCAP) * UDW;
            for (uint32_t i = (uint32_t)t; i < words; i += NT) (*lp(&L.ud[0][i])) = gp(gUd)[uo * UDW + i];
        }
    }
    BLOCK_SYNC();
    // what reaches chain slot `slot` at base jb + dj from the states that are not made in its own run
    // (acc: the other states are variable-length states whose cells of this block still sit in the accumulators -- the value of the
    //  base before is taken from there, as cellsOf makes it, so that this step need not wait for the cells)
    auto chainOthers = [&](int slot, int dj, int jb, int par, bool acc) __attribute__((always_inline)) {
        const int s2 = (*lp(&L.chS[slot])), j = jb + dj;
        double f = AUGX_NINF;
        int fa = 0xFF;
        if (s2 >= 0 && j >= 1 && j < n) {
            const double emi = (*lp(&L.sg[par][dj][(*lp(&L.chSgi[slot]))]));
            const int cc = clsAt(j), nd = (*lp(&L.chNd[slot]));
            double x[AUGX_MAX_ANC], m = AUGX_NINF;
            int fin = 0;
            for (int k2 = 0; k2 < AUGX_MAX_ANC; k2++) {
                x[k2] = AUGX_NINF;
                if (k2 < nd) {
                    const int a = (*lp(&L.chDead[slot][k2])), ai = (*lp(&L.chDeadAi[slot][k2]));
                    double pv;
                    if (acc && dj >= 1 && j - 1 >= 1) {
                        pv = (*lp(&L.cmax[par][dj - 1][a]));
                        if (FWD) pv = (*lp(&L.csum[par][dj - 1][a])) > 0ull ? pv + log((double)(*lp(&L.csum[par][dj - 1][a])) / FWD_FIX) : AUGX_NINF;
                    } else pv = ldsLoadD(&L.ring[(j - 1) & 63][a]);
                    if (pv > AUGX_NINF) x[k2] = pv + (trn(cc, s2, ai) + emi);
                    if (x[k2] > m) { m = x[k2]; fa = ai; } // (ascending ancestors, strict '>': the first of equals, as the reference)
                    fin += x[k2] > AUGX_NINF;
                }
            }
            f = m;
            if (FWD && fin > 1) {
                double sum = 0.0;
                for (int k2 = 0; k2 < AUGX_MAX_ANC; k2++) sum += x[k2] > AUGX_NINF ? exp(x[k2] - m) : 0.0;
                f = m + log(sum);
            }
        }
        (*lp(&L.oth[slot][dj])) = f;
        (*lp(&L.othAi[slot][dj])) = (uint8_t)fa;
    };
    // the chain state of `slot` over the bases of the block: itself (and the other chain states of its stage) from the base before
    auto chainRun = [&](int slot, int jb, int par) __attribute__((always_inline)) {
        const int s2 = (*lp(&L.chS[slot]));
        if (s2 < 0) return;
        const int na = (*lp(&L.chNa[slot])), sgi = (*lp(&L.chSgi[slot]));
        for (int dj = 0; dj < BLK; dj++) {
            const int j = jb + dj;
            if (j >= n || j < 1) continue;
            const int cc = clsAt(j);
            const double emi = (*lp(&L.sg[par][dj][sgi]));
            double f = (*lp(&L.oth[slot][dj])), f2 = AUGX_NINF; // (f2: the runner-up, for the near-tie flag)
            int fa = (*lp(&L.othAi[slot][dj]));
            for (int ai = 0; ai < na; ai++) {
                if (!(*lp(&L.chLive[slot][ai]))) continue;
                const int a = (*lp(&L.chAnc[slot][ai]));
                const double pv = ldsLoadD(&L.ring[(j - 1) & 63][a]);
                if (!(pv > AUGX_NINF)) continue;
                const double x = pv + (trn(cc, s2, ai) + emi);
                if (FWD) f = lse2(f, x);
                else if (x > f || (x == f && ai < fa)) { if (TIES) f2 = f; f = x; fa = ai; }
                else if (TIES && x > f2) f2 = x;
            }
            (*lp(&L.ring[j & 63][s2])) = f;
            // (near ties are being counted, dp.h: AUGX_NEAR_TIE: bit 6 of the back pointer says that the runner-up was that close)
            if (TIES && !FWD && fa < 8 && f2 > AUGX_NINF && f - f2 < AUGX_NEAR_TIE) fa |= 0x40;
            if (f > AUGX_NINF) { gp(M)[(int64_t)j * S + s2] = f; if (!FWD && BP) gp(BP)[(int64_t)j * S + s2] = (uint8_t)fa; }
        }
    };
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    uint64_t dpAcc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, dpLast = clock64();
#endif
    // the same when the state's only chain ancestor of its stage is the state itself: the previous value stays in a register
    auto chainRunSelf = [&](int slot, int s2, int selfAi, int sgi, int jb, int par) __attribute__((always_inline)) {
        double em[BLK], ot[BLK];
        int oa[BLK];
        _Pragma("unroll") for (int dj = 0; dj < BLK; dj++) { em[dj] = (*lp(&L.sg[par][dj][sgi])); ot[dj] = (*lp(&L.oth[slot][dj])); oa[dj] = (*lp(&L.othAi[slot][dj])); }
        const int jp = (jb >= 1 ? jb : 1) - 1;
        double prev = ldsLoadD(&L.ring[jp & 63][s2]);
        const double tSelf = (selfAi >= 0 && !multi) ? ldsLoadD(&L.tr[s2][selfAi]) : 0.0;
        _Pragma("unroll") for (int dj = 0; dj < BLK; dj++) {
            const int j = jb + dj;
            if (j >= n || j < 1) continue;
            double f = ot[dj], f2 = AUGX_NINF;
            int fa = oa[dj];
            if (selfAi >= 0 && prev > AUGX_NINF) {
                const double x = prev + ((multi ? trn(clsAt(j), s2, selfAi) : tSelf) + em[dj]);
                if (FWD) f = lse2(f, x);
                else if (x > f || (x == f && selfAi < fa)) { if (TIES) f2 = f; f = x; fa = selfAi; }
                else if (TIES) f2 = x;
            }
            if (TIES && !FWD && fa < 8 && f2 > AUGX_NINF && f - f2 < AUGX_NEAR_TIE) fa |= 0x40;
            (*lp(&L.ring[j & 63][s2])) = f;
            if (f > AUGX_NINF) { gp(M)[(int64_t)j * S + s2] = f; if (!FWD && BP) gp(BP)[(int64_t)j * S + s2] = (uint8_t)fa; }
            prev = f;
        }
    };
    bool lateAcc = true; // (uniform) every state that feeds a late chain state from outside its stage is a variable-length state of stage 2
    for (int slot = nEarly; slot < nCh; slot++) {
        const int s2 = chS[slot];
        for (int ai = 0; ai < T.n_anc[s2]; ai++) {
            const int a = T.anc[s2][ai], ka = T.kind[a];
            const bool live = isChainKind(ka) && !isEarlyChainKind(ka);
            if (!live && !((isItemKind(ka) && ka != AUGX_K_RTERMINAL) || isUtrExonKind(ka))) lateAcc = false;
        }
    }
    for (int b = 0; b < nBlocks; b++) {
        const int jb = b * BLK, par = b & 1;
        const int64_t gb = gb0 + b;
        DPROF(0);
        const bool redoBlock = nUv > 0 && jb + BLK > n - 1 && jb <= n - 1; // (the block of the last base: one of its cells is made twice, below)
        const uint64_t i0 = (*lp(&L.bOff[par]));
        const uint32_t cntAll = (*lp(&L.bCnt[par][0])), cntNonRT = (*lp(&L.bCnt[par][1]));
        auto candValue = [&](const Item &I, int &dj, int &s2) __attribute__((always_inline)) -> double {
            dj = (int)(I.kp >> (KEY_BITS + 7)); s2 = (int)((I.kp >> KEY_BITS) & 127);
            if (!(I.te > AUGX_NINF)) return AUGX_NINF;
            const int eop = (int)(I.kp & KEY_MASK) - KEY_BIAS;
#ifdef AUGX_EMU // (stage 2 relies on it: a predecessor inside the block is a fixed-lag state, made in stage 1)
            if (eop >= jb && T.kind[s2] != AUGX_K_RTERMINAL && !isFixedKind(T.kind[I.src & 127u]) && getenv("AUGX_EMU_CHECK_LAG"))
                fprintf(stderr, "emu: in-block predecessor that is not a fixed-lag state: state %d kind %d j %d eop %d jb %d src %d\n", s2, T.kind[s2], jb + dj, eop, jb, (int)(I.src & 127u));
#endif
            const double pv = denseAt<FWD>(L, M, S, eop, (int)(I.src & 127u), jb, BLK);
            return pv + I.te;
        };
        auto itemPass = [&](uint32_t lo2, uint32_t hi2, bool sumPass) __attribute__((always_inline)) {
            FOR_THREADS(t) {
                for (uint32_t it = lo2 + (uint32_t)(t - WAVE); t >= WAVE && it < hi2; it += NTW) {
                    int dj, s2;
                    const double v = candValue(ldItem(gItems + i0 + it), dj, s2);
                    if (!(v > AUGX_NINF)) continue;
                    if (!sumPass) ldsMaxD(&L.cmax[par][dj][s2], v);
                    else ldsAddU(&L.csum[par][dj][s2], (unsigned long long)(exp(v - (*lp(&L.cmax[par][dj][s2]))) * FWD_FIX));
                }
            }
        };
        // the UTR exon candidates of the block: its descriptors (kUtrDesc) are cut into units of 128 candidates, the units dealt to the
        // candidate wavefronts; within a unit the descriptor is the same for every lane, a lane takes two candidates and has the
        // loads of both in flight before it needs either (only >= 0: that one descriptor only)
        auto utrPass = [&](bool sumPass, int only, bool defer) __attribute__((always_inline)) {
            const uint32_t nd = (uint32_t)uni((int)(*lp(&L.uCnt[par])));
            const uint64_t uo = (*lp(&L.uOff[par]));
            FOR_WAVES(w) {
                if (w >= UW0) {
                    uint32_t wi = 0;
                    for (uint32_t d = only >= 0 ? (uint32_t)only : 0u; d < (only >= 0 ? (uint32_t)only + 1u : nd); d++) {
                        const int total = uni(d < (uint32_t)UDCAP ? ((const UDesc *)&L.ud[par][d * UDW])->total : B.ud[uo + d].total);
                        const int nun = (total + UH * WAVE - 1) / (UH * WAVE);
                        const int mine = (int)((uint32_t)(w - UW0 + (NWAVES - UW0) - (int)(wi % (NWAVES - UW0))) % (NWAVES - UW0)); // first unit of this descriptor that is this wavefront's
                        wi += (uint32_t)nun;
                        if (mine >= nun) continue;
                        const UDesc D = d < (uint32_t)UDCAP ? *(const UDesc *)&L.ud[par][d * UDW] : B.ud[uo + d];
                        const int s2 = D.s, dj = D.j - jb, cc = clsAt(D.j), na = (*lp(&L.nanc[s2]));
                        const USite *sites = (*lp(&L.siteTab[D.list]));
                        const int lsel = D.len;
                        for (int c = mine; c < nun; c += NWAVES - UW0) {
                            FOR_WLANES(t, w) {
                                int xi[UH], eop[UH], len[UH], lis[UH];
                                bool act[UH], tail3[UH], pre[UH];
                                int sPos[UH];
                                double sB[UH], sig[UH], lnLen[UH];
                                // the loads that depend on nothing: the site records of both candidates
                                _Pragma("unroll") for (int h = 0; h < UH; h++) {
                                    const int idx = c * UH * WAVE + h * WAVE + (t & 63);
                                    int li;
                                    act[h] = idx < total;
                                    utrCandIndex(D, idx, li, xi[h]);
                                    lis[h] = li;
                                    pre[h] = act[h] && xi[h] < 0 && li < D.nPre;
                                    sPos[h] = 0; sB[h] = AUGX_NINF;
                                    if (act[h] && xi[h] < 0 && !pre[h]) {
                                        USite e;
                                        e = ldUSite(sites + ((int64_t)D.i1 - 1 - li));
                                        sPos[h] = e.pos; sB[h] = utrSiteB(B, D, e);
                                    }
                                }
                                // the length terms (they depend on the site's position); the rare cases go to the list of stage 2
                                _Pragma("unroll") for (int h = 0; h < UH; h++) {
                                    lnLen[h] = AUGX_NINF;
                                    if (pre[h]) { // (evaluated by the descriptor kernel)
                                        sig[h] = lis[h] == 0 ? D.preTe[0] : lis[h] == 1 ? D.preTe[1] : D.preTe[2];
                                        eop[h] = lis[h] == 0 ? D.preEop[0] : lis[h] == 1 ? D.preEop[1] : D.preEop[2];
                                        lnLen[h] = 0.0;
                                        act[h] = sig[h] > AUGX_NINF;
                                    } else if (act[h]) {
                                        const int r = utrCandPre(UX, D, xi[h], sPos[h], sB[h], sig[h], len[h], tail3[h], eop[h], defer);
                                        act[h] = r == 1;
                                    }
                                    if (act[h] && !pre[h]) { // (utrLenAt, from the tables in LDS)
                                        const int ti = tail3[h] ? 8 : lsel;
                                        lnLen[h] = (len[h] >= 0 && len[h] <= (*lp(&L.lenMax[ti]))) ? gp((*lp(&L.lenTab[ti])))[len[h]] : AUGX_NINF;
                                    }
                                }
                                // the predecessors' values: every load in flight before the first is used
                                double pv[UH][4];
                                _Pragma("unroll") for (int h = 0; h < UH; h++)
                                    _Pragma("unroll") for (int ai = 0; ai < 4; ai++)
                                        pv[h][ai] = (act[h] && ai < na) ? denseAt<FWD>(L, M, S, eop[h], (*lp(&L.anc[s2][ai])), jb, BLK) : AUGX_NINF;
                                // every candidate of the unit belongs to one cell: the lane combines its own (largest / sum), then one atomic
                                double vmax = AUGX_NINF;
                                unsigned long long vsum = 0ull;
                                const double cm = sumPass ? (*lp(&L.cmax[par][dj][s2])) : 0.0;
                                _Pragma("unroll") for (int h = 0; h < UH; h++) {
                                    if (!act[h] || !(lnLen[h] > AUGX_NINF)) continue;
                                    const double te = sig[h] + lnLen[h];
                                    if (!(te > AUGX_NINF)) continue;
#ifdef AUGX_EMU
                                    if (eop[h] >= jb && only < 0 && getenv("AUGX_EMU_CHECK_LAG")) fprintf(stderr, "emu: in-block predecessor of a UTR exon: state %d j %d eop %d jb %d\n", s2, D.j, eop[h], jb);
#endif
                                    _Pragma("unroll") for (int ai = 0; ai < 4; ai++) {
                                        if (ai >= na || !(pv[h][ai] > AUGX_NINF)) continue;
                                        const double v = pv[h][ai] + (trn(cc, s2, ai) + te);
                                        if (!sumPass) vmax = v > vmax ? v : vmax;
                                        else vsum += (unsigned long long)(exp(v - cm) * FWD_FIX);
                                    }
                                    for (int ai = 4; ai < na; ai++) { // (no UTR exon state of the reference's models has more than four ancestors)
                                        const double pw = denseAt<FWD>(L, M, S, eop[h], (*lp(&L.anc[s2][ai])), jb, BLK);
                                        if (!(pw > AUGX_NINF)) continue;
                                        const double v = pw + (trn(cc, s2, ai) + te);
                                        if (!sumPass) vmax = v > vmax ? v : vmax;
                                        else vsum += (unsigned long long)(exp(v - cm) * FWD_FIX);
                                    }
                                }
                                if (!sumPass) { if (vmax > AUGX_NINF) ldsMaxD(&L.cmax[par][dj][s2], vmax); }
                                else if (vsum) ldsAddU(&L.csum[par][dj][s2], vsum);
                            }
                        }
                    }
                }
            }
        };
        auto cellsOf = [&](int t, int kindMask) __attribute__((always_inline)) { // (thread t >= WAVE) the cells of the states whose cellKind is in the mask
            for (int i = t - WAVE; i < BLK * SPX; i += NTW) {
                const int dj = i / SPX, s2 = i % SPX, j = jb + dj;
                if (j >= 1 && j < n && ((kindMask >> (*lp(&L.cellKind[s2]))) & 1)) {
                    double f = (*lp(&L.cmax[par][dj][s2]));
                    if (FWD) f = (*lp(&L.csum[par][dj][s2])) > 0ull ? f + log((double)(*lp(&L.csum[par][dj][s2])) / FWD_FIX) : AUGX_NINF;
                    (*lp(&L.ring[j & 63][s2])) = f;
                    if (f > AUGX_NINF) gp(M)[(int64_t)j * S + s2] = f;
                }
            }
        };
        // ---- 1: fixed-lag states; the UTR exon candidates (their predecessors end before the block); the accumulators, offsets,
        //         signal records and descriptors of the NEXT block
        FOR_THREADS(t) {
            for (int i = t; t < A1T && i < BLK * SPX; i += A1T) { (*lp(&L.cmax[par ^ 1][i / SPX][i % SPX])) = AUGX_NINF; (*lp(&L.csum[par ^ 1][i / SPX][i % SPX])) = 0ull; }
            if (t == A1T - 1 && b + 1 < nBlocks) {
                (*lp(&L.bOff[par ^ 1])) = gp(gBlkOff)[(gb + 1) * 2 + 1]; (*lp(&L.bCnt[par ^ 1][0])) = gp(gBlkCnt)[(gb + 1) * 2 + 1]; (*lp(&L.bCnt[par ^ 1][1])) = gp(gBlkSplit)[(gb + 1) * 3 + 2];
            }
            if (b + 1 < nBlocks && t >= A1T - BLK * NSIG && t < A1T) {
                const int i = t - (A1T - BLK * NSIG), dj = i / NSIG, j = jb + BLK + dj;
                (*lp(&L.sg[par ^ 1][dj][i % NSIG])) = j < n ? gp(gSig)[(int64_t)j * NSIG + i % NSIG] : AUGX_NINF;
            }
            if (nUv > 0 && b + 1 < nBlocks && t < A1T) { // the next block's descriptors
                const uint64_t uo = gp(gUdOff)[gb + 1];
                const uint32_t uc = gp(gUdCnt)[gb + 1];
                if (t == 0) { (*lp(&L.uOff[par ^ 1])) = uo; (*lp(&L.uCnt[par ^ 1])) = uc; }
                const uint32_t words = (uc < (uint32_t)UDCAP ? uc : (uint32_t)UDCAP) * UDW;
                for (uint32_t i = (uint32_t)t; i < words; i += A1T) (*lp(&L.ud[par ^ 1][i])) = gp(gUd)[uo * UDW + i];
            }
            _Pragma("unroll") for (int r = 0; r < FR; r++)
            if (fS2[r][TI] >= 0) {
                const int s2 = fS2[r][TI], dj = (t + r * A1T) % BLK, j = jb + dj;
                if (j >= 1 && j < n) {
                    const int lag = fLag[r][TI];
                    const double emi = (*lp(&L.sg[par][dj][fSg[r][TI]]));
                    double f = AUGX_NINF;
                    int fa = 0xFF;
                    if (j - lag >= 0 && emi > AUGX_NINF) {
                        const int cc = clsAt(j), na = (*lp(&L.nanc[s2]));
                        for (int ai = 0; ai < na; ai++) {
                            const double pv = denseAt<FWD>(L, M, S, j - lag, (*lp(&L.anc[s2][ai])), jb, BLK);
                            if (!(pv > AUGX_NINF)) continue;
                            const double x = pv + (trn(cc, s2, ai) + emi);
                            if (FWD) f = lse2(f, x);
                            else if (x > f) { f = x; fa = ai; }
                        }
                    }
                    (*lp(&L.ring[j & 63][s2])) = f;
                    if (f > AUGX_NINF) { gp(M)[(int64_t)j * S + s2] = f; if (!FWD && BP) gp(BP)[(int64_t)j * S + s2] = (uint8_t)fa; }
                }
            }
        }
        DPROF(1);
        if (nUv > 0) utrPass(false, -1, false);
        BLOCK_SYNC();
        DPROF(2);
        // ---- 2: the records of kCand but RTERMINAL (their predecessors: earlier blocks, or fixed-lag states of this one); the early chain
        //         states (geometric introns: fed by the fixed-lag states of the base before and by themselves)
        itemPass(0, cntNonRT, false);
        FOR_THREADS(t) { if (t >= WAVE && t - WAVE < nEarly * BLK) chainOthers((t - WAVE) / BLK, (t - WAVE) % BLK, jb, par, false); }
        BLOCK_SYNC();
        DPROF(3);
        if (FWD) { itemPass(0, cntNonRT, true); if (nUv > 0) utrPass(true, -1, false); BLOCK_SYNC(); }
        DPROF(4);
        // ---- 3: the cells of the variable-length states; the early chain states over the block
        FOR_THREADS(t) {
            if (t >= WAVE) cellsOf(t, (1 << 1) | (1 << 3));
                if (t < nEarly) { if (TX(cFast)) chainRunSelf(t, TX(cS2), TX(cSelf), TX(cSgi), jb, par); else chainRun(t, jb, par); }
            // (what reaches the late chain states, where it can be read from the accumulators: beside the cells instead of after them)
            if (lateAcc && !redoBlock && t >= NT - (nCh - nEarly) * BLK) { const int u = t - (NT - (nCh - nEarly) * BLK); chainOthers(nEarly + u / BLK, u % BLK, jb, par, true); }
        }
        BLOCK_SYNC();
        DPROF(5);
        // the right-truncated 3' UTR exon at the last base of the piece may begin anywhere up to that base (src/utrmodel.cc:880-884):
        // its predecessors of this very block exist only now -- the cell is made once more, from all of them
        if (nUv > 0 && jb + BLK > n - 1 && jb <= n - 1) {
            const uint32_t nd = (*lp(&L.uCnt[par]));
            const uint64_t uo = (*lp(&L.uOff[par]));
            for (uint32_t d = 0; d < nd; d++) {
                const UDesc D = d < (uint32_t)UDCAP ? *(const UDesc *)&L.ud[par][d * UDW] : B.ud[uo + d];
                if (D.kind != AUGX_K_UTR3SINGLE || D.j != n - 1 || D.total == 0) continue;
                const int s2 = D.s;
                FOR_THREADS(t) { if (t == 0) { (*lp(&L.cmax[par][n - 1 - jb][s2])) = AUGX_NINF; (*lp(&L.csum[par][n - 1 - jb][s2])) = 0ull; } }
                BLOCK_SYNC();
                utrPass(false, (int)d, false);
                BLOCK_SYNC();
                if (FWD) { utrPass(true, (int)d, false); BLOCK_SYNC(); }
                FOR_THREADS(t) {
                    if (t == 0) {
                        double f = (*lp(&L.cmax[par][n - 1 - jb][s2]));
                        if (FWD) f = (*lp(&L.csum[par][n - 1 - jb][s2])) > 0ull ? f + log((double)(*lp(&L.csum[par][n - 1 - jb][s2])) / FWD_FIX) : AUGX_NINF;
                        (*lp(&L.ring[(n - 1) & 63][s2])) = f;
                        gp(M)[(int64_t)(n - 1) * S + s2] = f;
                    }
                }
                BLOCK_SYNC();
            }
        }
        DPROF(6);
        // ---- 4: late chain states (intergenic, UTR introns: fed by the exon cells of the base before and by themselves)
        if (!(lateAcc && !redoBlock)) {
            FOR_THREADS(t) { if (t >= WAVE && t - WAVE < (nCh - nEarly) * BLK) chainOthers(nEarly + (t - WAVE) / BLK, (t - WAVE) % BLK, jb, par, false); }
            BLOCK_SYNC();
        }
        FOR_THREADS(t) { if (t >= nEarly && t < nCh) { if (TX(cFast)) chainRunSelf(t, TX(cS2), TX(cSelf), TX(cSgi), jb, par); else chainRun(t, jb, par); } }
        BLOCK_SYNC();
        DPROF(7);
        // ---- 5: reverse terminal exons (they may start from a cell of their own block)
        itemPass(cntNonRT, cntAll, false);
        BLOCK_SYNC();
        if (FWD) { itemPass(cntNonRT, cntAll, true); BLOCK_SYNC(); }
        DPROF(8);
        FOR_THREADS(t) { if (t >= WAVE) cellsOf(t, 1 << 2); }
        BLOCK_SYNC();
        DPROF(9);
        // the columns of this block reach HBM before any later block reads them from there (the ring covers 64 bases)
        if (((b + 1) * BLK) % 32 == 0) BLOCK_GLOBAL_SYNC();
        DPROF(10);
    }
#if !defined(AUGX_EMU) && defined(AUGX_PROFILE)
    if (B.prof && threadIdx.x == WAVE) for (int k = 0; k < 12; k++) B.prof[(int64_t)p * 56 + k] = dpAcc[k];
#endif
    BLOCK_SYNC();
    FOR_THREADS(t) { // termination (reference NAMGene::getViterbiPath, src/namgene.cc:442-462; getSampledPath :385-392)
        if (t == 0) {
            double tot = AUGX_NINF;
            int fin = -1;
            for (int i = 0; i < S; i++) {
                const double tl = termKind == 0 ? T.ln_term[i] : (i == synch ? 0.0 : AUGX_NINF);
                const double v = (*lp(&L.ring[(n - 1) & 63][i])) + tl;
                if (!(v > AUGX_NINF)) continue;
                if (FWD) tot = lse2(tot, v);
                else if (v > tot) { tot = v; fin = i; }
            }
            if (FWD) B.lnFwd[p] = tot;
            else { B.lnv[p] = tot; B.finalState[p] = fin; B.status[p] = fin < 0 ? AUGX_E_NOPATH : fabs(tot) < AUGX_EXACT_LIMIT ? 0 : AUGX_E_RANGE; }
        }
    }
}

// =================================================================================================
// back-tracking over the dense matrix (reference NAMGene::getViterbiPath, src/namgene.cc:467-506): chain and fixed-lag states
// follow their stored ancestor; a variable-length state takes the arg-max over its candidates again (records of kCand, or the
// UTR site list), same additions, ties to the larger predecessor end, then to the ancestor of lower index
// =================================================================================================
AUGX_KFN void denseBacktracePiece(const DevTables &T, const BatchView &B, int p) {
    const int n = B.len[p], S = T.S;
    const int64_t o = B.off[p];
    const int64_t po = pathOff(B, p), cap = pathCap(B, p);
    const double *M = B.cells + (o + 1) * S;
    const uint8_t *BP = B.bpD + (o + 1) * S;
    int state = B.finalState[p], base = n - 1, count = 0, nearTies = 0;
    bool overflow = false;
    if (state < 0 || B.status[p] != 0) { FOR_LANES(l) { if (l == 0) { B.pathCount[p] = 0; if (B.nearTie) B.nearTie[p] = 0; } } return; }
    const int dssWhole = T.Ds + 2 + T.De, assLag = T.As + 2 + T.Ae + T.U;
    UCtx UX(T, B, p);
    auto colOf = [&](int eop) { return eop > 0 ? eop : 0; };
    while (base > 0) {
        const int kind = T.kind[state];
        int eop, ai;
        if (isChainKind(kind)) {
            int selfAi = -1;
            for (int i = 0; i < T.n_anc[state]; i++) if (T.anc[state][i] == state) selfAi = i;
            int cur = base, w = 0xFF;
            for (;;) { // the first base <= cur whose predecessor is not the state itself: 4 bases per lane, 256 per step
                LV(int, flag); LV(int, wv); LV(int, hit);
                FOR_LANES(l) {
                    LX(flag) = 0; LX(wv) = -1; LX(hit) = 4;
                    for (int k = 3; k >= 0; k--) {
                        const int q = cur - 4 * l - k;
                        const int raw = q >= 1 ? (int)BP[(int64_t)q * S + state] : -1;
                        const int ww = raw < 0 || raw >= 0xFE ? raw : (raw & 0x3F); // (bit 6: the decision of that cell was a near tie)
                        if (q < 1 || ww != selfAi) { LX(flag) = 1; LX(wv) = ww; LX(hit) = k; }
                    }
                }
                const int first = waveFirstTrue(flag);
                if (first < WAVE) {
#ifdef AUGX_EMU
                    const int hk = hit[first], hw = wv[first];
#else
                    const int hk = __shfl(hit[0], first, 64), hw = __shfl(wv[0], first, 64);
#endif
                    cur -= 4 * first + hk;
                    w = cur >= 1 ? hw : 0xFF;
                    break;
                }
                cur -= 4 * WAVE;
            }
            if (B.nearTie) { // the cells of this run whose decision (stay / come in from another state) was a near tie
                for (int q0 = cur < 1 ? 1 : cur; q0 <= base; q0 += WAVE) {
                    LV(int, fl);
                    FOR_LANES(l) { const int q = q0 + l; const int raw = q <= base ? (int)BP[(int64_t)q * S + state] : 0xFF; LX(fl) = (raw < 0xFE && (raw & 0x40)) ? 1 : 0; }
                    nearTies += waveCount(fl);
                }
            }
            if (cur < 1) { eop = 0; ai = -1; }
            else if (w == 0xFE) { eop = 0; ai = -1; } // (a piece of N only: intergenic from the first base)
            else { eop = cur - 1; ai = w; }
        } else if (isFixedKind(kind)) {
            ai = BP[(int64_t)base * S + state];
            if (ai == 0xFF) { overflow = true; break; }
            eop = base - ((kind == AUGX_K_LONGDSS || kind == AUGX_K_RLONGDSS) ? dssWhole : (kind == AUGX_K_LONGASS || kind == AUGX_K_RLONGASS) ? assLag : T.dStateLen);
        } else if (isUtrExonKind(kind)) {
            UDesc D;
            utrDescribe(UX, state, base, D);
            const int cc = UX.clsAt(base);
            Best best{AUGX_NINF, -2147483647, -1};
            double runnerUp = AUGX_NINF; // (pass 1, only when near ties are counted: the largest (predecessor end, ancestor) that is not the winner)
            for (int pass = 0; pass < (B.nearTie ? 2 : 1); pass++)
            for (int c0 = 0; c0 < D.total; c0 += WAVE) {
                LV(double, cv); LV(int, ck); LV(int, ca);
                FOR_LANES(l) {
                    LX(cv) = AUGX_NINF; LX(ck) = -2147483647; LX(ca) = -1;
                    double te; int e2;
                    if (c0 + l < D.total && utrCand(UX, D, c0 + l, te, e2))
                        for (int a2 = 0; a2 < T.n_anc[state]; a2++) { // (ascending, strict '>': the ancestor of lower index wins a tie)
                            const double pv = M[(int64_t)colOf(e2) * S + T.anc[state][a2]];
                            if (!(pv > AUGX_NINF)) continue;
                            if (pass == 1 && e2 + KEY_BIAS == best.key && a2 == best.aux) continue; // (the winner itself)
                            const double v = pv + (lnT(T, cc, T.anc[state][a2], state) + te);
                            if (v > LX(cv)) { LX(cv) = v; LX(ck) = e2 + KEY_BIAS; LX(ca) = a2; }
                        }
                }
                const Best b2 = waveArgMax(cv, ck, ca);
                if (pass == 0) { if (better(b2.v, b2.key, best.v, best.key)) best = b2; }
                else if (b2.v > runnerUp) runnerUp = b2.v;
            }
            if (B.nearTie && runnerUp > AUGX_NINF && best.v - runnerUp < AUGX_NEAR_TIE && best.v != runnerUp) nearTies++; // (an exact tie is decided by the reference's own rule, the same in both)
            if (!(best.v > AUGX_NINF)) { overflow = true; break; }
            ai = best.aux; eop = best.key - KEY_BIAS;
        } else {
            const int blkSz = B.blk;
            const int64_t gb = o / blkSz + base / blkSz;
            const uint64_t i0 = B.blkOff[gb * 2 + 1];
            const uint32_t cnt = B.blkCnt[gb * 2 + 1], pid = (uint32_t)(((base % blkSz) << 7) | state);
            Best best{AUGX_NINF, -2147483647, -1};
            double runnerUp = AUGX_NINF;
            for (int pass = 0; pass < (B.nearTie ? 2 : 1); pass++)
            for (uint32_t c0 = 0; c0 < cnt; c0 += WAVE) {
                LV(double, cv); LV(int, ck); LV(int, ca);
                FOR_LANES(l) {
                    LX(cv) = AUGX_NINF; LX(ck) = -2147483647; LX(ca) = -1;
                    const uint32_t it = c0 + (uint32_t)l;
                    if (it < cnt) {
                        const Item I = B.items[i0 + it];
                        if ((I.kp >> KEY_BITS) == pid && I.te > AUGX_NINF) {
                            const int e2 = (int)(I.kp & KEY_MASK) - KEY_BIAS, a = (int)(I.src & 127u);
                            const double pv = M[(int64_t)colOf(e2) * S + a];
                            if (pv > AUGX_NINF) {
                                int a2 = 0;
                                for (int i = 0; i < T.n_anc[state]; i++) if (T.anc[state][i] == a) a2 = i;
                                // (key: the predecessor end, then the ancestor of LOWER index among equals)
                                LX(cv) = pv + I.te; LX(ck) = (e2 + KEY_BIAS) * AUGX_MAX_ANC + (AUGX_MAX_ANC - 1 - a2); LX(ca) = a2;
                                if (pass == 1 && LX(ck) == best.key) LX(cv) = AUGX_NINF; // (the winner itself)
                            }
                        }
                    }
                }
                const Best b2 = waveArgMax(cv, ck, ca);
                if (pass == 0) { if (better(b2.v, b2.key, best.v, best.key)) best = b2; }
                else if (b2.v > runnerUp) runnerUp = b2.v;
            }
            if (B.nearTie && runnerUp > AUGX_NINF && best.v - runnerUp < AUGX_NEAR_TIE && best.v != runnerUp) nearTies++; // (an exact tie is decided by the reference's own rule, the same in both)
            if (!(best.v > AUGX_NINF)) { overflow = true; break; }
            ai = best.aux; eop = best.key / AUGX_MAX_ANC - KEY_BIAS;
        }
        if (count >= cap) { overflow = true; break; }
        FOR_LANES(l) { if (l == 0) { int32_t *r = B.pathRec + (po + count) * 3; r[0] = eop + 1; r[1] = base; r[2] = state; } }
        count++;
        base = eop;
        if (ai < 0 || ai >= T.n_anc[state]) { if (base > 0) overflow = true; break; }
        state = T.anc[state][ai];
    }
    FOR_LANES(l) { if (l == 0) { B.pathCount[p] = count; if (B.nearTie) B.nearTie[p] = nearTies; if (overflow) B.status[p] = AUGX_E_HIP; } }
}

} // namespace dev
} // namespace augx
