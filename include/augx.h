/*
 * augx.h -- C ABI of the MI355X-native AUGUSTUS decode path (libaugx.so).
 *
 * The reference (Gaius-Augustus/Augustus v3.5.0) has no FFI; its seam for this path is the C++ call pair
 *     NAMGene::viterbiAndForward(const char* dna)      reference include/namgene.hh:91,  src/namgene.cc:168-365
 *     NAMGene::getViterbiPath(const char* dna, ...)    reference include/namgene.hh:41,  src/namgene.cc:432-510
 * driven per "piece" by NAMGene::doViterbiPiecewise     reference src/namgene.cc:516-676
 * over the per-state virtual StateModel::viterbiForwardAndSampling (reference include/statemodel.hh:76-77).
 * Every entry point below names the reference interface it replaces.  Plain pointers and sizes only.
 *
 * All functions return 0 on success and a negative AUGX_E_* code on error; the message is available from
 * augx_last_error() (thread-local, NUL-terminated).  No entry point ever falls back to a CPU decode:
 * without a usable HIP device augx_decoder_create() fails with AUGX_E_NODEVICE.
 */
#ifndef AUGX_H
#define AUGX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define AUGX_MAX_STATES 80
#define AUGX_MAX_ANC 8
#define AUGX_MAX_CLASSES 16
/* Markov-chain content sums are accumulated in fixed point (ln p * 2^40 rounded to int64, wrap-around
 * uint64 adds): exactly associative, so device scans of any shape give bit-identical prefix differences. */
/* Every ln term of the model tables is a multiple of 2^-AUGX_Q_BITS (rounded once when the model is loaded): all sums on the
 * decode path stay below 2^(52-AUGX_Q_BITS) = 2^21 * 2 in magnitude (pieces shorter than 3 Mbp), hence every fp64 addition
 * is EXACT -- associative and shift-invariant like integer arithmetic, with -inf for probability 0 kept for free. */
#define AUGX_Q_BITS 31
#define AUGX_FX_SHIFT 40
#define AUGX_FX_SCALE 1099511627776.0          /* 2^40  */
#define AUGX_FX_INV (1.0 / 1099511627776.0)    /* 2^-40 */

enum {
    AUGX_OK = 0,
    AUGX_E_ARG = -1,        /* bad argument                                                            */
    AUGX_E_CONFIG = -2,     /* species / model configuration could not be read (ProjectError analogue)  */
    AUGX_E_NODEVICE = -3,   /* no HIP device / kernel image: the product path has no CPU fallback       */
    AUGX_E_HIP = -4,        /* HIP runtime error                                                        */
    AUGX_E_UNSUPPORTED = -5,/* model feature outside the implemented hot path (fails loudly)            */
    AUGX_E_NOPATH = -6,     /* "No feasible path found in HMM" (reference src/namgene.cc:455-457)       */
    AUGX_E_NOMEM = -7,      /* device memory exhausted (decode fewer bases per batch)                   */
    AUGX_E_RANGE = -8       /* |ln V| of a piece left the range in which every fp64 addition is exact (AUGX_Q_BITS): the
                               piece is too improbable for its length -- lower --maxDNAPieceSize                  */
};

/* state kinds of the GHMM (derived from the reference's StateType, include/types.hh:492-512) */
enum {
    AUGX_K_IGENIC = 0,
    AUGX_K_SINGLE, AUGX_K_INITIAL, AUGX_K_INTERNAL, AUGX_K_TERMINAL,           /* forward coding exons  */
    AUGX_K_RSINGLE, AUGX_K_RINITIAL, AUGX_K_RINTERNAL, AUGX_K_RTERMINAL,       /* reverse coding exons  */
    AUGX_K_LESSD, AUGX_K_LONGDSS, AUGX_K_EQUALD, AUGX_K_GEOMETRIC, AUGX_K_LONGASS,      /* fwd intron  */
    AUGX_K_RLESSD, AUGX_K_RLONGDSS, AUGX_K_REQUALD, AUGX_K_RGEOMETRIC, AUGX_K_RLONGASS, /* rev intron  */
    /* untranslated regions (reference UtrModel, src/utrmodel.cc; states_shadow_utr.cfg): kind = AUGX_K_UTR5SINGLE + (type - utr5single)
       on the forward strand, AUGX_K_RUTR5SINGLE + (type - rutr5single) on the reverse strand */
    AUGX_K_UTR5SINGLE, AUGX_K_UTR5INIT, AUGX_K_UTR5INTRON, AUGX_K_UTR5INTRONVAR, AUGX_K_UTR5INTERNAL, AUGX_K_UTR5TERM,
    AUGX_K_UTR3SINGLE, AUGX_K_UTR3INIT, AUGX_K_UTR3INTRON, AUGX_K_UTR3INTRONVAR, AUGX_K_UTR3INTERNAL, AUGX_K_UTR3TERM,
    AUGX_K_RUTR5SINGLE, AUGX_K_RUTR5INIT, AUGX_K_RUTR5INTRON, AUGX_K_RUTR5INTRONVAR, AUGX_K_RUTR5INTERNAL, AUGX_K_RUTR5TERM,
    AUGX_K_RUTR3SINGLE, AUGX_K_RUTR3INIT, AUGX_K_RUTR3INTRON, AUGX_K_RUTR3INTRONVAR, AUGX_K_RUTR3INTERNAL, AUGX_K_RUTR3TERM
};

/*
 * Flat, immutable model tables.  Everything the reference keeps in static class members after
 * StateModel::readAllParameters() (reference src/statemodel.cc:232-238; exon :604-792, intron :295-415,
 * igenic :150-225) and NAMGene::readTransAndInitProbs (src/namgene.cc:1318-1392), as natural logarithms
 * in fp64 (-inf for probability 0).  Pattern indices are base-4 numbers, a=0 c=1 g=2 t=3, first base most
 * significant (reference Seq2Int, include/geneticcode.hh:163-241).  NP = 4^(k+1).
 */
typedef struct augx_tables {
    int32_t S;                      /* number of states (47 for human/fly without UTR, 71 with, 48 with two intergenic states) */
    int32_t n_classes;              /* GC-content classes (decomp_num_steps)                            */
    int32_t k;                      /* order of the exon/intron/igenic Markov chains (all equal)        */
    int32_t W, U, As, Ae, Ds, De;   /* trans_init_window, ass_upwindow_size, ass_start/end, dss_start/end */
    int32_t Li, Le;                 /* init_coding_len, et_coding_len                                   */
    int32_t d;                      /* /IntronModel/d                                                   */
    int32_t max_exon_len, min_exon_len, min_coding_len;
    int32_t tis_n, tis_k, ass_n, ass_k;   /* motif widths / orders                                      */
    int32_t tis_nbins;              /* >0: TIS probabilities are binned (not human)                     */
    int32_t tis_mem;                /* /ExonModel/tis_motif_memory (reference src/exonmodel.cc:1319)     */
    int32_t synch_state;
    int32_t gc_win;                 /* GCwinsize                                                        */
    int32_t state_type[AUGX_MAX_STATES];  /* reference StateType id (for names / GFF logic)             */
    int32_t state_kind[AUGX_MAX_STATES];  /* AUGX_K_*                                                   */
    int32_t state_win[AUGX_MAX_STATES];   /* stateReadingFrames[type] (reference src/types.cc:174-189)  */
    int32_t reachable[AUGX_MAX_STATES];
    double ln_init[AUGX_MAX_STATES];      /* [Initial] of the transition file                           */
    double ln_term[AUGX_MAX_STATES];      /* [Terminal]                                                 */
    const double *ln_trans;         /* [n_classes][S][S]  ln t(a->s) after the per-class intron rewrite
                                       (reference IntronModel::updateToLocalGCEach, src/intronmodel.cc:439-488) */
    /* ---- per GC class, class-major ---- */
    const double *ig_emi;           /* [C][NP]       igenic emission (intron table when tieIgenicIntron) */
    const double *ig_short;         /* [C][k+1][NP]  ln of the short-pattern ratio used for positions <= k
                                       (reference IGenicModel::emiProbUnderModel, src/igenicmodel.cc:342-356) */
    const double *in_emi;           /* [C][4^(k_in+1)] intron emission (k_in = k for every shipped species but one, see k_in)  */
    const double *ex_emi;           /* [C][3][NP]    exon content, frame-dependent                      */
    const double *ex_init;          /* [C][3][NP]    initial content                                    */
    const double *ex_et;            /* [C][3][NP]    exon-terminal content                              */
    const double *ex_pls;           /* [C][k+1][3][NP] joint l-mer probabilities P_ls (only 4^(l+1) used)*/
    const double *tis_motif;        /* [C][tis_n][4^(tis_k+1)]                                          */
    const double *ass_motif;        /* [C][ass_n][4^(ass_k+1)]                                          */
    const double *tis_bin_bounds;   /* [C][tis_nbins-1] (linear probabilities) or NULL                  */
    const double *tis_bin_ln;       /* [C][tis_nbins]                                                   */
    /* ---- class independent ---- */
    const double *ass_pat;          /* [4^(As+Ae)]  ln of the (binned) ASS pattern probability           */
    const double *dss_pat;          /* [4^(Ds+De)]  ln of the (binned) DSS pattern probability (dss_gc: twice as long, see there) */
    double ass_pat_invalid;         /* ln(0.001 * 0.25^(As+Ae)), reference src/intronmodel.cc:1179        */
    const double *len_intron;       /* [d+1]                                                            */
    const double *len_single;       /* [max_exon_len+1]  ln(3*P(len)) as used in notEndPartEmiProb       */
    const double *len_initial;
    const double *len_internal;
    const double *len_terminal;
    double ln_startcodon[64];       /* ln startCodonProb, -inf if not a start codon                      */
    double ln_stop_ochre, ln_stop_amber, ln_stop_opal;   /* taa, tag, tga                                */
    double ln_quarter;              /* ln 0.25  */
    double ln_n_coding;             /* ln probNinCoding (0.23) */
    double ln4;                     /* ln 4 */
    /* GC-class decomposition (reference ContentDecomposition, src/motif.cc:458-505) */
    double gc_zus[AUGX_MAX_CLASSES][4];   /* ra, rc, rg, rt of the class centres                         */
    double gc_weight_matrix[16];
    int32_t gc_weighing_type;       /* 1 equal, 2 gcContentClasses, 3 multiNormalKernel                  */
    int32_t n_anc[AUGX_MAX_STATES];
    int32_t anc[AUGX_MAX_STATES][AUGX_MAX_ANC];  /* ancestors in ascending state index (tie-break order) */
    /* soft-masking (reference SequenceFeatureCollection::prepare, src/extrinsicinfo.cc:1696-1724): every lower-case
       base is a nonexonpart hint of source RM; igenic and intron states get its bonus per covered base */
    int32_t softmasking;            /* 0/1: --softmasking                                                */
    double ln_soft_bonus;           /* ln(bonus), 1.15 in config/extrinsic/extrinsic.cfg                  */
    /* ---- untranslated regions (--UTR=on; reference UtrModel::readAllParameters, src/utrmodel.cc:540-696, the
     *      constants of UtrModel::init :149-260 and Constant::init, src/types.cc:303,407).  utr == 0: none of this is set ---- */
    int32_t utr;                    /* 0/1                                                               */
    int32_t tss_upwin;              /* /Constant/tss_upwindow_size                                        */
    int32_t tss_start, tss_end, tata_start, tata_end, d_tss_tata_min, d_tss_tata_max;
    int32_t d_polyasig_cleavage, aataaa_boxlen, tts_spacing;
    int32_t utr_max_exon_len, utr_max3single, utr_max3term;   /* maxexonlength, max3singlelength, max3termlength */
    int32_t tssup_k;                /* order of the TSS-upstream-window chain                             */
    int32_t tss_n, tss_k, tsstata_n, tsstata_k, tata_n, tata_k, tts_n, tts_k;   /* motif widths / orders   */
    const double *utr5init_emi;     /* [C][NP]  5' single/initial exon content, mixed with the intron table by
                                       utr5patternweight (src/utrmodel.cc:681-688)                         */
    const double *utr5_emi;         /* [C][NP]  5' internal/terminal exon content                         */
    const double *utr3_emi;         /* [C][NP]  3' exon content                                           */
    const double *tssup_emi;        /* [C][4^(tssup_k+1)]                                                  */
    const double *tss_motif, *tsstata_motif, *tata_motif, *tts_motif;   /* [C][n][4^(k+1)]                 */
    const double *aataaa;           /* [4^boxlen] ln(aataaa_probs * prob_polya), -inf where the file has no entry */
    double ln_tts_rand;             /* ln((1 - prob_polya) / 4^boxlen), src/utrmodel.cc:1853,1883           */
    const double *len5_single, *len5_initial, *len5_internal, *len5_terminal;   /* [utr_max_exon_len+1]    */
    const double *len3_single;      /* [utr_max3single+1] */
    const double *len3_initial, *len3_internal;                                 /* [utr_max_exon_len+1]    */
    const double *len3_terminal;    /* [utr_max3term+1]   */
    const double *tail5_single;     /* [utr_max_exon_len+1] tail probabilities (truncated UTR at the piece start) */
    const double *tail3_single;     /* [utr_max3single+1]   */
    double ln2;                     /* ln 2 (negative-length corrections pow(2.0, .), src/utrmodel.cc:1180) */
    /* /IntronModel/allow_dss_consensus_gc (Constant::dss_gc_allowed, include/geneticcode.hh:47-54): a donor site may read gc as well as
     * gt; dss_pat then holds a second half, [4^(Ds+De) ..), for the gc sites: ln of the (binned) pattern probability times
     * non_gt_dss_prob (src/intronmodel.cc:1232-1239) */
    int dss_gc;
    /* Markov order of the three UTR exon content tables (UtrModel::k: the `k` lines of [EMISSION-5INITIAL] / [EMISSION-5] /
     * [EMISSION-3] in the species' utr_probs file, src/utrmodel.cc:617-655): utr5init_emi / utr5_emi / utr3_emi are
     * [C][4^(utr_k+1)].  Where it differs from k (chlamydomonas, chlamy2011, culex: 3 against 4) the reference (a) mixes entry i of
     * the UTR table with entry i of the INTRON table of order k (:681-688) and (b) scores a base of a utr5intron / utr3intron
     * state with the intron pattern that begins utr_k bases before it, i.e. ends k - utr_k bases after it (:1255-1262,1389-1396);
     * both are reproduced */
    int utr_k;
    /* --translation_table (GeneticCode::chooseTranslationTable, src/geneticcode.cc:146-170): which of taa (bit 0), tag (bit 1),
     * tga (bit 2) end a reading frame -- 7 for the standard code, 4 for table 6 (ciliates: taa, tag read glutamine).  The open
     * reading frames, the stop of single / terminal exons and the stop-codon veto of short introns follow the table, not the
     * {ochre,amber,opal}prob values (src/exonmodel.cc:204-219).  A table with a stop codon other than these three is refused:
     * the reference throws at the first such codon that ends a gene (src/exonmodel.cc:1284-1292) */
    int stop_mask;
    /* bit c: codon c (index a=0 c=1 g=2 t=3, first base most significant) may start a gene under the translation table
     * (GeneticCode::isStartcodon; table 1: atg, ctg, ttg) -- whatever probability the species gives it.  Read by the end gate of
     * the 5' UTR states next to the start codon (src/utrmodel.cc:1075-1078) */
    unsigned long long start_mask;
    /* Markov order of the intron content table where it is not k (tetrahymena: 3 beside 4; IntronModel::k, the `k` line of the
     * [EMISSION] section of the species' intron file, src/intronmodel.cc:356-372): in_emi is [C][4^(k_in+1)].  Without UTR states only */
    int k_in;
} augx_tables;

typedef struct augx_model augx_model;     /* host-side immutable model: tables + option values          */
typedef struct augx_decoder augx_decoder; /* device-side context bound to one HIP device + stream       */

/* one unit of work = one "piece" (reference src/namgene.cc:584-603) */
/* values of augx_piece.init_kind / term_kind (reference src/namgene.cc:584-603) */
#define AUGX_INIT_FILE 0
#define AUGX_INIT_SYNCH 1
#define AUGX_TERM_FILE 0
#define AUGX_TERM_SYNCH 1

typedef struct augx_piece {
    const char *seq;      /* ASCII nucleotides, any case; non-acgt = invalid (HOST pointer)               */
    int64_t len;
    int32_t init_kind;    /* 0: [Initial] probs of the transition file; 1: synch state only (interior cut) */
    int32_t term_kind;    /* 0: [Terminal] probs;                        1: synch state only               */
} augx_piece;

/* == State(begin,end,type) of the reference (include/gene.hh:101-135), 0-based HMM-state coordinates */
typedef struct augx_state {
    int32_t begin, end;
    int16_t state;        /* state index in the model                                                     */
    int16_t type;         /* reference StateType id                                                        */
} augx_state;

typedef struct augx_path {
    augx_state *states;   /* 5'->3'; runs of single-base igenic / geometric states are merged             */
    int32_t n_states;
    int32_t status;       /* 0 or AUGX_E_NOPATH                                                           */
    double ln_viterbi;    /* ln of StatePath::pathemiProb (reference src/namgene.cc:462)                  */
} augx_path;

/* ---- model (replaces Properties::init + Constant::init + NAMGene() + StateModel::readAllParameters,
 *      reference src/augustus.cc:111-176) ---- */
int augx_model_load(const char *config_path, const char *species, int n_opts, const char *const *opt_names,
                    const char *const *opt_values, augx_model **out);
const augx_tables *augx_model_tables(const augx_model *m);
const char *augx_model_option(const augx_model *m, const char *name); /* NULL if unset */
void augx_model_destroy(augx_model *m);

/* ---- decoder (device) ---- */
int augx_device_count(void);              /* visible HIP devices (0 without a GPU: there is no CPU decode path)           */
int augx_decoder_create(const augx_model *m, int device, augx_decoder **out);
void augx_decoder_destroy(augx_decoder *d);
/* how many decoders (streams) decode on this device at the same time (default 1).  The trellis kernel cuts pieces into
 * segments so that every compute unit has a workgroup; a decoder that shares the device plans for its share of them. */
int augx_decoder_set_share(augx_decoder *d, int n_decoders_on_device);
/* bases one batch should hold at most: what fits the free device memory (about 1.5 KB per base), capped at 128 Mbp */
/* Pieces with several GC classes: the reference's short-intron content cache (SnippetProbs, src/statemodel.cc:312-342) is not
 * emptied at a class step, so within 2 d bases after one an interior may be scored in chunks of different classes.  With
 * exact = 1 (the default; environment AUGX_EXACT_MULTICLASS=0 turns it off) the cache is replayed after a first trellis run and
 * the run repeated with the rebuilt terms: every Viterbi variable is then the reference's to 1e-9 there as well -- and the path:
 * a randomised soak found a record where the optimal path depends on it.  exact = 0 saves the second trellis run on batches
 * with such pieces.  The forward algorithm (augx_batch_forward) always replays it.
 * Models with UTR states: two more call-history caches of the reference that only UTR states read are replayed with it --
 * tssProbsPlus (a forward TSS window is scored with the class current when a 5' UTR state first asks for it,
 * src/utrmodel.cc:748-790,1788-1790) and the memo of IntronModel::aSSProb (first asker, emptied beyond 1000 sites,
 * src/intronmodel.cc:1120-1135,1182-1186; device/assmemo.h). */
int augx_decoder_set_exact(augx_decoder *d, int exact);
int augx_decoder_exact(const augx_decoder *d); /* the current setting (1 / 0) */
/* Near ties.  Every model term is rounded once to 2^-31 (AUGX_Q_BITS), which is what makes the decode exact and order-free; two
 * alternative candidates of a cell whose scores differ by less than ~2e-7 in ln may therefore be decided the other way by the
 * reference, whose own rounding is finer (seen once in 315 randomised runs).  With counting on (also: AUGX_TIMING or
 * AUGX_NEAR_TIES=1 in the environment when the decoder is created) the trellis flags the cells of the single-base chain states where staying
 * and the best way in from another state were within 2e-7 of each other, and the back-trace counts those ON THE CHOSEN PATH together with
 * the cells of the variable-length states on it whose runner-up candidate lies within 2e-7 of the winner; augx_decoder_near_ties returns the
 * sum over the paths fetched so far (and the number of pieces with at least one).  0 = no decision of the path was that close.  Counting
 * runs a build of the kernels of its own (about 1.5 % slower); off by default. */
int augx_decoder_count_near_ties(augx_decoder *d, int on);
int64_t augx_decoder_near_ties(const augx_decoder *d, int64_t *pieces /* may be NULL */);
/* host buffers kept between sampled pieces (forward matrices, at most 12 GB) are released when the last decoder is destroyed, or here */
void augx_release_host_pools(void);
int64_t augx_decoder_batch_capacity(augx_decoder *d);
/* the same when the forward matrix (8 S bytes per base) is made on top of the decode: what augx_decode_sampled packs into a batch
 * (reference: NAMGene::viterbiAndForward keeps both matrices of a piece, src/namgene.cc:168-365) */
int64_t augx_decoder_sampled_capacity(augx_decoder *d);

/* replaces viterbiAndForward + getViterbiPath for a batch of independent pieces */
int augx_decode_batch(augx_decoder *d, const augx_piece *pieces, int n, augx_path *out /* array[n] */);
void augx_path_free(augx_path *p);

/* ---- multi-GPU: pieces (and the cut finder's exam windows) are independent, so they shard over devices with no
 *      exchange step.  One decoder per device; pieces are assigned longest-first to the least loaded
 *      device (augx_partition_lpt, host only), each device is driven by its own host thread in batches bounded by its
 *      free memory, and the results land in input order in out[0..n).  This is the fan-out of the piece loop of
 *      NAMGene::doViterbiPiecewise (reference src/namgene.cc:575-676); the reference itself is single-threaded. ---- */
int augx_partition_lpt(const int64_t *lens, int n, int n_bins, int32_t *bin_of /* [n] */);
/* The TSS window that begins at base 0 of a sequence, forward (out[0]) and reverse (out[1]), ln; -inf without UTR states.
 * The reference keeps entry 0 of its tssProbsPlus / tssProbsMinus from sequence to sequence while the sequences it decodes -- pieces
 * and the exam windows of its cut finder, in its order -- keep ONE length: the entry is neither cleared at a class step
 * (UtrModel::updateToLocalGC clears [from, to) with from = 1, src/utrmodel.cc:779-781) nor re-allocated (initAlgorithms, :744-747).
 * A piece that follows such a sequence is therefore decoded with the value that sequence computed for ITS base 0:
 * augx_tss0 computes the pair for a sequence, augx_tss0_override hands it to the piece whose `seq` pointer is given (v = NULL:
 * forget it) -- read by every augx_batch_create that gets that pointer.  augx_main does this for the pieces of a run. */
int augx_tss0(const augx_model *m, const char *seq, int64_t len, double *out /* [2] */);
int augx_tss0_override(const char *seq, const double *v /* [2] or NULL */);
int augx_decode_sharded(augx_decoder *const *decs, int n_dec, const augx_piece *pieces, int n, augx_path *out /* array[n] */);

/* ---- the cut finder: where a record longer than maxDNAPieceSize is cut into pieces.  Replaces NAMGene::getNextCutEndPoint /
 *      tryFindCutEndPoint as driven by the piece loop of doViterbiPiecewise (reference src/namgene.cc:575-603, 973-1210): per
 *      round an exam window at the end of the range is decoded and the record is cut at the centre of the largest intergenic
 *      region of its path (second try with a window twice as long, then the fall-backs of :1100-1133).  The chain of cuts is
 *      serial in the reference; here the windows are decoded through `fn` in batches that run ahead of the chain (a scout decode
 *      of the record forecasts where the cuts will fall; every cut still comes from the decode of exactly the window the
 *      reference decodes).  `fn` decodes n independent pieces (e.g. augx_decode_batch / augx_decode_sharded on the caller's
 *      decoders) and returns 0 or an AUGX_E_* code.  Result: the pieces of all records in order, with the initial / terminal
 *      kinds the piece loop gives them (:584-603); free with augx_cuts_free. ---- */
typedef int (*augx_decode_fn)(void *user, const augx_piece *pieces, int n, augx_path *out /* array[n] */);
typedef struct augx_cut {
    int32_t record;
    int32_t status;      /* 0, or the status of the exam window of this record that could not be decoded (then the record has no further pieces) */
    int64_t begin, end;  /* 0-based, inclusive */
    int32_t init_kind, term_kind;
} augx_cut;
typedef struct augx_cut_stats { int32_t scout_tiles, batches, windows_decoded, windows_used; } augx_cut_stats;
int augx_find_cuts(const augx_model *m, int n_records, const char *const *seqs, const int64_t *lens, augx_decode_fn fn, void *user,
                   int scout /* -1: decide from the chain length, 0: one window per round, 1: decode ahead */, augx_cut **out, int *n_out,
                   augx_cut_stats *stats /* may be NULL */);
void augx_cuts_free(augx_cut *c);

/* ---- device-resident batch interface (bench / multi-GPU driver): the batch is staged once, decode can be
 *      repeated and timed with the inputs resident in HBM ---- */
typedef struct augx_batch augx_batch;
int augx_batch_create(augx_decoder *d, const augx_piece *pieces, int n, augx_batch **out); /* H2D copy  */
int augx_batch_decode(augx_decoder *d, augx_batch *b);            /* all kernels, async on the stream   */
int augx_batch_sync(augx_decoder *d);
int augx_batch_paths(augx_decoder *d, augx_batch *b, augx_path *out /* array[n] */);      /* D2H + unpack */
int augx_batch_kernel_ms(augx_decoder *d, augx_batch *b, float *prep_ms, float *trellis_ms, float *back_ms);
/* test hook: copy the dense ln V[j][s] matrix (len*S doubles, -inf = absent) of piece i to host; only
 * valid on a decoder created with AUGX_DEBUG_CELLS=1 in the environment */
int augx_batch_cells(augx_decoder *d, augx_batch *b, int piece, double *out);
/* forward algorithm of a decoded batch (reference NAMGene::viterbiAndForward with needForwardTable, src/namgene.cc:168-365,
 * the per-state `fwdsum`s): the dense ln F matrix stays on the device for the posterior sampling; the second call copies the
 * len*S matrix of one piece (-inf = absent) and ln P(sequence) to the host (tests; the executable's --sample > 0 goes through
 * augx_decode_sampled) */
int augx_batch_forward(augx_decoder *d, augx_batch *b);
int augx_batch_forward_cells(augx_decoder *d, augx_batch *b, int piece, double *out, double *ln_p);
/* posterior sampling (reference NAMGene::getSampledPath, src/namgene.cc:367-426, OptionsList::sample, src/vitmatrix.cc:295-320):
 * n_samples state paths of one piece of a batch whose forward matrix has been computed, drawn backwards from the last column;
 * every step lists its options in the reference's order, sorts them by probability and draws with
 * rand() / RAND_MAX * sum * 0.99999.  The reference draws from glibc's rand() (never seeded: seed 1), one call per step, over
 * the whole run: augx_rand is that generator (the TYPE_3 additive-feedback generator of glibc, restated), kept by the caller
 * across pieces in input order (the executable: one generator per run, augx_decode_sampled). */
typedef struct augx_rand augx_rand;
augx_rand *augx_rand_create(unsigned seed);
int augx_rand_next(augx_rand *r);               /* == rand() of glibc after srand(seed) */
void augx_rand_skip(augx_rand *r, int64_t n);   /* n draws spent unseen (a sampled path draws once per base of an intergenic run)  */
void augx_rand_destroy(augx_rand *r);
int augx_batch_sample(augx_decoder *d, augx_batch *b, int piece, int n_samples, augx_rand *r, augx_path *out /* array[n_samples] */);
/* decode + sample n pieces on n_dec devices: out[i] = the Viterbi path of piece i, samples[i * n_samples + k] = its k-th sampled
 * path; the draws are taken from r in input order (piece 0's first), whatever device a piece ran on */
int augx_decode_sampled(augx_decoder *const *decs, int n_dec, const augx_piece *pieces, int n, int n_samples, augx_rand *r,
                        augx_path *out /* array[n] */, augx_path *samples /* array[n * n_samples] */);
void augx_batch_destroy(augx_batch *b);

/* ---- whole-program driver (replaces main(), reference src/augustus.cc:94-248): same argv as `augustus`,
 *      GFF on `out_fd`, diagnostics on `err_fd`; returns the process exit code ---- */
int augx_main(int argc, const char *const *argv);

/* ---- gene-structure + GFF stage on its own (replaces StatePath::projectOntoGeneSequence + filterGenePrediction +
 *      printGeneList, reference src/gene.cc:394-700,2465-2524,3071-3120) for one record decoded as one piece ---- */
int augx_format_gff(const augx_model *m, const char *name, const char *seq, int64_t len, const augx_state *states,
                    int n_states, int first_gene_id, char *out, int64_t out_cap, int *n_genes);
/* the same with n_samples sampled paths next to the Viterbi path (--sample = n_samples + 1): transcripts of the sample are
 * united with the Viterbi transcripts, the score columns carry the posterior probabilities (reference NAMGene::findGenes,
 * src/namgene.cc:795-905) */
int augx_format_gff_sampled(const augx_model *m, const char *name, const char *seq, int64_t len, const augx_state *states,
                            int n_states, int n_samples, const augx_state *const *sample_states, const int *sample_n,
                            int first_gene_id, char *out, int64_t out_cap, int *n_genes);

/* ---- the ordered gather of a sharded run (reference NAMGene::doViterbiPiecewise, src/namgene.cc:526,626-650: gene ids are
 *      numbered over the whole run in input order): the pieces of n_records records, decoded on any device and handed over
 *      in any order, become the prediction part of the `augustus` output (block headers of src/augustus.cc:395-398). ---- */
typedef struct augx_piece_result {
    int32_t record;            /* index of the FASTA record the piece belongs to                                 */
    int32_t status;            /* 0 or an AUGX_E_* code (the record's error)                                     */
    int64_t begin, end;        /* piece = bases [begin, end] of the record                                       */
    const augx_state *states;  /* decoded path in piece coordinates                                              */
    int32_t n_states;
} augx_piece_result;
int augx_format_records(const augx_model *m, int n_records, const char *const *names, const char *const *seqs,
                        const int64_t *lens, int n_pieces, const augx_piece_result *pieces, char *out, int64_t out_cap);

const char *augx_last_error(void);
const char *augx_version(void);

#ifdef __cplusplus
}
#endif
#endif
