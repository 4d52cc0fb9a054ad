//! Safe wrappers over libmsi.  Error policy: every failure is an `Err(GpuError)`; the
//! caller (milli) logs it and falls back to its own arroy/hannoy/fst path — a search never fails because
//! the accelerator did (same policy as `search/hybrid.rs:326-336` on embedding failure).
pub mod sys;

use std::ffi::CStr;
use std::ptr::{self, NonNull};

use roaring::RoaringBitmap;

#[derive(Debug)]
pub struct GpuError { pub status: i32, pub message: String }

fn check(status: i32) -> Result<(), GpuError> {
    if status == sys::MSI_OK { return Ok(()); }
    let message = unsafe { CStr::from_ptr(sys::msi_last_error()) }.to_string_lossy().into_owned();
    Err(GpuError { status, message })
}

/// One per (process, GPU).  Objects created on it keep it alive inside the library, so drop order is free.
pub struct GpuContext(NonNull<sys::msi_ctx>);
unsafe impl Send for GpuContext {}
unsafe impl Sync for GpuContext {} // every entry point is thread-safe; calls serialise on the stream locks
impl GpuContext {
    /// `device < 0`: `$LOCAL_RANK` if set, else device 0.  Fails unless a gfx950 device is present.
    pub fn new(device: i32) -> Result<Self, GpuError> {
        // the struct layouts of sys.rs are those of ABI 2 (msi_search_params: geo_strategy, geo_cache_size, index_view)
        let abi = unsafe { sys::msi_abi_version() };
        if abi != sys::MSI_ABI_VERSION {
            return Err(GpuError { status: sys::MSI_E_UNSUPPORTED, message: format!("libmsi ABI {abi} != {}", sys::MSI_ABI_VERSION) });
        }
        let mut p = ptr::null_mut();
        check(unsafe { sys::msi_ctx_create(device, &mut p) })?;
        Ok(Self(NonNull::new(p).expect("msi_ctx_create returned NULL with MSI_OK")))
    }
}
impl Drop for GpuContext { fn drop(&mut self) { unsafe { sys::msi_ctx_destroy(self.0.as_ptr()) } } }

/// Dense words (bit i = docid i) of a `RoaringBitmap` — the boundary form of `filter`/`candidates`.
pub fn dense_words(bitmap: &RoaringBitmap) -> Vec<u64> {
    let n = bitmap.max().map_or(0, |m| m as usize / 64 + 1);
    let mut w = vec![0u64; n];
    for id in bitmap { w[id as usize >> 6] |= 1u64 << (id & 63); }
    w
}

/// One arroy/hannoy store (one (embedder, store id) pair) resident in HBM.
pub struct GpuVectorStore { h: NonNull<sys::msi_vs>, dim: usize }
unsafe impl Send for GpuVectorStore {}
unsafe impl Sync for GpuVectorStore {}
impl GpuVectorStore {
    pub fn new(ctx: &GpuContext, dim: usize, bf16: bool) -> Result<Self, GpuError> {
        let mut p = ptr::null_mut();
        let storage = if bf16 { sys::MSI_VS_BF16 } else { sys::MSI_VS_F32 };
        check(unsafe { sys::msi_vs_create_typed(ctx.0.as_ptr(), dim as u32, storage, &mut p) })?;
        let this = Self { h: NonNull::new(p).unwrap(), dim };
        // concurrent spawn_blocking searches share HBM sweeps (up to 16/32/48 queries each)
        check(unsafe { sys::msi_vs_set_microbatch(this.h.as_ptr(), 200) })?;
        Ok(this)
    }
    /// `docids` strictly ascending (arroy's item iteration order), `rows` row-major `[n][dim]`.
    pub fn upload(&mut self, docids: &[u32], rows: &[f32]) -> Result<(), GpuError> {
        assert_eq!(rows.len(), docids.len() * self.dim);
        check(unsafe { sys::msi_vs_upload(self.h.as_ptr(), docids.as_ptr(), rows.as_ptr(), docids.len() as u64) })
    }
    /// A committed update (update/new/indexer/write.rs:65-74,157: `del_item` / `add_item` per document): `remove` leaves
    /// the store, `add` enters it (an existing docid is replaced); both strictly ascending.  Only the delta crosses PCIe.
    pub fn update(&mut self, remove: &[u32], add_docids: &[u32], add_rows: &[f32]) -> Result<(), GpuError> {
        assert_eq!(add_rows.len(), add_docids.len() * self.dim);
        check(unsafe { sys::msi_vs_update(self.h.as_ptr(), remove.as_ptr(), remove.len() as u64, add_docids.as_ptr(),
                                          add_rows.as_ptr(), add_docids.len() as u64) })
    }
    /// `VectorStore::nns_by_vector` for this store (crates/milli/src/vector/store.rs:638-675).
    pub fn nns_by_vector(&self, vector: &[f32], limit: usize, filter: Option<&RoaringBitmap>,
                         cancel: Option<&std::sync::atomic::AtomicI32>) -> Result<Vec<(u32, f32)>, GpuError> {
        assert_eq!(vector.len(), self.dim);
        let words = filter.map(dense_words);
        let (fp, fb) = words.as_ref().map_or((ptr::null(), 0), |w| (w.as_ptr(), w.len() as u64 * 64));
        let (mut ids, mut dist, mut n) = (vec![0u32; limit], vec![0f32; limit], 0u32);
        let cp = cancel.map_or(ptr::null(), |c| c.as_ptr() as *const i32);
        check(unsafe { sys::msi_vs_search(self.h.as_ptr(), vector.as_ptr(), 1, limit as u32, fp, fb, cp,
                                          ids.as_mut_ptr(), dist.as_mut_ptr(), &mut n) })?;
        Ok(ids.into_iter().zip(dist).take(n as usize).collect())
    }
    /// `VectorStore::nns_by_item` (store.rs:615-637): `None` when the item has no vector in this store.
    pub fn nns_by_item(&self, item: u32, limit: usize, filter: Option<&RoaringBitmap>)
        -> Result<Option<Vec<(u32, f32)>>, GpuError> {
        let words = filter.map(dense_words);
        let (fp, fb) = words.as_ref().map_or((ptr::null(), 0), |w| (w.as_ptr(), w.len() as u64 * 64));
        let (mut ids, mut dist, mut n, mut found) = (vec![0u32; limit], vec![0f32; limit], 0u32, 0i32);
        check(unsafe { sys::msi_vs_search_by_item(self.h.as_ptr(), item, limit as u32, fp, fb, ids.as_mut_ptr(),
                                                  dist.as_mut_ptr(), &mut n, &mut found) })?;
        Ok((found != 0).then(|| ids.into_iter().zip(dist).take(n as usize).collect()))
    }
}
impl Drop for GpuVectorStore { fn drop(&mut self) { unsafe { sys::msi_vs_destroy(self.h.as_ptr()) } } }

/// The words FST staged flat in HBM (`fst.stream()` order = byte-lexicographic).
pub struct GpuDictionary { h: NonNull<sys::msi_dict>, concat: Vec<u8>, offsets: Vec<u32> }
unsafe impl Send for GpuDictionary {}
unsafe impl Sync for GpuDictionary {}
impl GpuDictionary {
    pub fn from_sorted_words<'a>(ctx: &GpuContext, words: impl Iterator<Item = &'a [u8]>) -> Result<Self, GpuError> {
        let (mut concat, mut offsets) = (Vec::new(), vec![0u32]);
        for w in words { concat.extend_from_slice(w); offsets.push(concat.len() as u32); }
        let mut p = ptr::null_mut();
        check(unsafe { sys::msi_dict_create(ctx.0.as_ptr(), concat.as_ptr(), offsets.as_ptr(),
                                            offsets.len() as u32 - 1, &mut p) })?;
        let this = Self { h: NonNull::new(p).unwrap(), concat, offsets };
        check(unsafe { sys::msi_dict_set_microbatch(this.h.as_ptr(), 200, 256) })?;
        Ok(this)
    }
    /// Straight from `index.words_fst(rtxn)?.as_fst().as_bytes()` (index.rs:1225-1243): the library decodes the
    /// `fst` 0.4 bytes itself (checksum-verified), so no key is streamed through Rust.  The flat copy kept here
    /// only resolves indices back to words; an `Err` (older fst version, corrupt bytes) means "stream it".
    pub fn from_fst_bytes(ctx: &GpuContext, fst: &[u8]) -> Result<Self, GpuError> {
        let (mut n, mut nb) = (0u32, 0u64);
        check(unsafe { sys::msi_fst_decode(fst.as_ptr(), fst.len(), 0, ptr::null_mut(), 0, ptr::null_mut(), 0, &mut n, &mut nb) })?;
        let (mut concat, mut offsets) = (vec![0u8; nb as usize], vec![0u32; n as usize + 1]);
        check(unsafe { sys::msi_fst_decode(fst.as_ptr(), fst.len(), 1 /* MSI_FST_SKIP_CHECKSUM */, concat.as_mut_ptr(), nb,
                                           offsets.as_mut_ptr(), n, &mut n, &mut nb) })?;
        let mut p = ptr::null_mut();
        check(unsafe { sys::msi_dict_create(ctx.0.as_ptr(), concat.as_ptr(), offsets.as_ptr(), n, &mut p) })?;
        let this = Self { h: NonNull::new(p).unwrap(), concat, offsets };
        check(unsafe { sys::msi_dict_set_microbatch(this.h.as_ptr(), 200, 256) })?;
        Ok(this)
    }
    pub fn word(&self, idx: u32) -> &str {
        let (a, b) = (self.offsets[idx as usize] as usize, self.offsets[idx as usize + 1] as usize);
        std::str::from_utf8(&self.concat[a..b]).expect("dictionary words are UTF-8")
    }
    /// `find_one_typo_derivations` (max_typos = 1) / `find_one_two_typo_derivations` (2):
    /// (one-typo indices, two-typo indices), in fst stream order.
    pub fn derivations(&self, word: &str, max_typos: u8, is_prefix: bool) -> Result<(Vec<u32>, Vec<u32>), GpuError> {
        const CAP1: usize = 150; // MAX_ONE_TYPO_COUNT, search/new/limits.rs:7
        const CAP2: usize = 50;  // MAX_TWO_TYPOS_COUNT, limits.rs:9
        let q = sys::msi_typo_query { word: word.as_ptr(), len: word.len() as u32, max_typos, is_prefix: is_prefix as u8, _pad: 0 };
        let (mut one, mut two, mut n1, mut n2) = (vec![0u32; CAP1], vec![0u32; CAP2], 0u32, 0u32);
        check(unsafe { sys::msi_dict_lookup(self.h.as_ptr(), &q, 1, CAP1 as u32, CAP2 as u32, one.as_mut_ptr(),
                                            &mut n1, two.as_mut_ptr(), &mut n2) })?;
        one.truncate(n1 as usize);
        two.truncate(n2 as usize);
        Ok((one, two))
    }
}
impl Drop for GpuDictionary { fn drop(&mut self) { unsafe { sys::msi_dict_destroy(self.h.as_ptr()) } } }

/// A pool of dense docid sets in HBM (slots are the handles).
pub struct GpuDocidSets { h: NonNull<sys::msi_bits>, pub n_slots: u32 }
unsafe impl Send for GpuDocidSets {}
impl GpuDocidSets {
    pub fn new(ctx: &GpuContext, n_docs: u64, n_slots: u32) -> Result<Self, GpuError> {
        let mut p = ptr::null_mut();
        check(unsafe { sys::msi_bits_create(ctx.0.as_ptr(), n_docs, n_slots, &mut p) })?;
        Ok(Self { h: NonNull::new(p).unwrap(), n_slots })
    }
    /// slot := decode of a `CboRoaringBitmapCodec` value, exactly the bytes LMDB returns.
    pub fn set_from_cbo(&mut self, slot: u32, bytes: &[u8]) -> Result<(), GpuError> {
        check(unsafe { sys::msi_bits_set_from_cbo(self.h.as_ptr(), slot, bytes.as_ptr(), bytes.len()) })
    }
    pub fn raw(&self) -> *mut sys::msi_bits { self.h.as_ptr() }
}
impl Drop for GpuDocidSets { fn drop(&mut self) { unsafe { sys::msi_bits_destroy(self.h.as_ptr()) } } }

/// One hit of the keyword leg with the `ScoreDetails` of the two rules.
#[derive(Debug, Clone, Copy)]
pub struct KeywordHit { pub docid: u32, pub matching_words: u32, pub max_matching_words: u32, pub typo_count: u32, pub max_typo_count: u32 }

/// What `msi_keyword_search` needs from the index (LMDB gets of `db_cache.rs`), as a trait the shim
/// implements on `SearchContext`.  The byte slices are the stored `CboRoaringBitmap` values.
pub trait PostingSource {
    fn word_docids(&mut self, word: &str, original: bool) -> Option<&[u8]>;
    fn word_pair_proximity_docids(&mut self, proximity: u8, left: &str, right: &str) -> Option<&[u8]>;
    fn is_exact_word(&mut self, word: &str) -> bool;
}

// The three trampolines turn the trait object back into the C vtable (user = *mut &mut dyn PostingSource).
unsafe extern "C" fn tramp_word(user: *mut std::ffi::c_void, w: *const u8, n: u32, original: i32,
                                out: *mut *const u8, out_n: *mut usize) -> i32 {
    let src = &mut **(user as *mut &mut dyn PostingSource);
    let Ok(word) = std::str::from_utf8(std::slice::from_raw_parts(w, n as usize)) else { return -1 };
    match src.word_docids(word, original != 0) {
        Some(b) => { *out = b.as_ptr(); *out_n = b.len(); }
        None => { *out_n = 0; }
    }
    0
}
unsafe extern "C" fn tramp_pair(user: *mut std::ffi::c_void, prox: u32, l: *const u8, ln: u32, r: *const u8, rn: u32,
                                out: *mut *const u8, out_n: *mut usize) -> i32 {
    let src = &mut **(user as *mut &mut dyn PostingSource);
    let (Ok(l), Ok(r)) = (std::str::from_utf8(std::slice::from_raw_parts(l, ln as usize)),
                          std::str::from_utf8(std::slice::from_raw_parts(r, rn as usize))) else { return -1 };
    match src.word_pair_proximity_docids(prox as u8, l, r) {
        Some(b) => { *out = b.as_ptr(); *out_n = b.len(); }
        None => { *out_n = 0; }
    }
    0
}
unsafe extern "C" fn tramp_exact(user: *mut std::ffi::c_void, w: *const u8, n: u32) -> i32 {
    let src = &mut **(user as *mut &mut dyn PostingSource);
    std::str::from_utf8(std::slice::from_raw_parts(w, n as usize)).map_or(0, |w| src.is_exact_word(w) as i32)
}

/// The keyword leg for the rule list `[Words, Typo]` (bucket_sort.rs:23-343 over those two rules).
#[allow(clippy::too_many_arguments)]
pub fn keyword_search(dict: &GpuDictionary, sets: &mut GpuDocidSets, source: &mut dyn PostingSource,
                      words: &[&str], last_is_prefix: bool, all_terms: bool, use_typo: bool,
                      min_one: u32, min_two: u32, authorize_typos: bool, universe: Option<&[u8]>,
                      from: usize, length: usize) -> Result<(Vec<KeywordHit>, u64), GpuError> {
    let tokens: Vec<_> = words.iter().enumerate().map(|(i, w)| sys::msi_query_token {
        word: w.as_ptr(), len: w.len() as u32, is_prefix: (last_is_prefix && i + 1 == words.len()) as u32 }).collect();
    let params = sys::msi_keyword_params {
        authorize_typos: authorize_typos as u32, min_word_len_one_typo: min_one, min_word_len_two_typos: min_two,
        strategy: if all_terms { sys::MSI_TERMS_ALL } else { sys::MSI_TERMS_LAST }, use_typo: use_typo as i32,
        from: from as u32, length: length as u32 };
    let mut src_ref: &mut dyn PostingSource = source;
    let vt = sys::msi_index_vtable { user: &mut src_ref as *mut _ as *mut _, word_docids: Some(tramp_word),
        word_pair_proximity_docids: Some(tramp_pair), is_exact_word: Some(tramp_exact),
        word_fid_docids: None, word_position_docids: None, word_fids: None, word_positions: None,
        field_id_word_count_docids: None, word_prefix_docids: None, word_prefix_fid_docids: None,
        word_prefix_position_docids: None, word_prefix_pair_proximity_docids: None, word_prefix_fids: None,
        word_prefix_positions: None, synonyms: None, exact_words_with_prefix: None };
    let (mut ids, mut mw, mut tc, mut mt) = (vec![0u32; length], vec![0u32; length], vec![0u32; length], vec![0u32; length]);
    let (mut n, mut cand) = (0u32, 0u64);
    let (up, ul) = universe.map_or((ptr::null(), 0), |u| (u.as_ptr(), u.len()));
    check(unsafe { sys::msi_keyword_search(dict.h.as_ptr(), sets.raw(), &vt, tokens.as_ptr(), tokens.len() as u32,
                                           &params, up, ul, ids.as_mut_ptr(), mw.as_mut_ptr(), tc.as_mut_ptr(),
                                           mt.as_mut_ptr(), &mut n, &mut cand) })?;
    let hits = (0..n as usize).map(|i| KeywordHit { docid: ids[i], matching_words: mw[i],
        max_matching_words: words.len() as u32, typo_count: tc[i], max_typo_count: mt[i] }).collect();
    Ok((hits, cand))
}

// ------------------------------------------------------------------------------------------------------
// The keyword leg with every graph-based ranking rule (msi_keyword_search_ranked)
// ------------------------------------------------------------------------------------------------------

/// Everything the ranked search reads from the index: the LMDB gets of `search/new/db_cache.rs`, each returning
/// the stored `CboRoaringBitmap` bytes untouched.  `SearchContext` implements it in the shim.
pub trait RankingSource: PostingSource {
    fn word_fid_docids(&mut self, word: &str, fid: u16) -> Option<&[u8]>;
    fn word_position_docids(&mut self, word: &str, position: u16) -> Option<&[u8]>;
    fn word_fids(&mut self, word: &str) -> Vec<u16>;
    fn word_positions(&mut self, word: &str) -> Vec<u16>;
    fn field_id_word_count_docids(&mut self, fid: u16, count: u8) -> Option<&[u8]>;
    /// `word_prefix_docids` (+ `exact_word_prefix_docids` when `original`): every stored value goes to `push`;
    /// returns how many there were (0 = the prefix is not a key, the engine enumerates derivations instead).
    fn word_prefix_docids(&mut self, prefix: &str, original: bool, push: &mut dyn FnMut(&[u8])) -> usize;
    fn word_prefix_fid_docids(&mut self, prefix: &str, fid: u16, push: &mut dyn FnMut(&[u8])) -> usize;
    fn word_prefix_position_docids(&mut self, prefix: &str, position: u16, push: &mut dyn FnMut(&[u8])) -> usize;
    /// `prefix_iter` over `word_pair_proximity_docids` with the key (proximity, word1, prefix2…), db_cache.rs:451-520.
    fn word_prefix_pair_proximity_docids(&mut self, proximity: u8, word1: &str, prefix2: &str,
                                         push: &mut dyn FnMut(&[u8])) -> usize;
    fn word_prefix_fids(&mut self, prefix: &str) -> Vec<u16>;
    fn word_prefix_positions(&mut self, prefix: &str) -> Vec<u16>;
    /// `index.synonyms.get(words)`, each synonym tokenised.
    fn synonyms(&mut self, words: &[&str]) -> Vec<Vec<String>>;
    /// keys of `exact_word_docids` with the prefix, in key order (compute_derivations.rs:40-73)
    fn exact_words_with_prefix(&mut self, prefix: &str) -> Vec<String>;
}

/// One located query term of `located_query_terms_from_tokens` (parse_query.rs:28-202).
pub struct LocatedTerm<'a> {
    /// one word, or the words of a quoted phrase (`None` = a stop word inside the phrase)
    pub words: Vec<Option<&'a str>>,
    pub is_phrase: bool,
    pub is_negative: bool,
    pub is_prefix: bool,
    pub positions: std::ops::RangeInclusive<u16>,
}

/// `ScoreDetails` of the keyword rules (score_details.rs:10-27) as the engine reports them.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum RankedScore {
    Words { matching_words: u32, max_matching_words: u32 },
    Typo { typo_count: u32, max_typo_count: u32 },
    Proximity { rank: u32, max_rank: u32 },
    Fid { rank: u32, max_rank: u32 },
    Position { rank: u32, max_rank: u32 },
    ExactAttribute { rank: u32 },           // 3 ExactMatch, 2 MatchesStart, 1 NoExactMatch
    ExactWords { matching_words: u32, max_matching_words: u32 },
    Skipped,
    /// `ScoreDetails::Sort`: `rule` = index into `RankedSearch::order_keys`, `key` = the bucket's order key
    /// (`u32::MAX`: the documents without a value, `value: Null`)
    Sort { rule: u32, key: u32 },
    /// `ScoreDetails::GeoSort`: `rule` = index into `RankedSearch::geo_rules`; `value` = the `_geo` point of document
    /// `first_docid` (`geo_value`), `None` for `u32::MAX` (the documents without a point)
    GeoSort { rule: u32, first_docid: u32 },
}

/// `milli::TermsMatchingStrategy` (crates/milli/src/search/mod.rs:538-556)
#[derive(Clone, Copy, PartialEq, Eq)]
pub enum TermsStrategy { Last, All, Frequency }
impl TermsStrategy {
    fn as_msi(self) -> i32 {
        match self { Self::Last => sys::MSI_TERMS_LAST, Self::All => sys::MSI_TERMS_ALL, Self::Frequency => sys::MSI_TERMS_FREQUENCY }
    }
}

pub struct RankedSearch<'a> {
    pub criteria: &'a [i32],                // index.criteria() as sys::MSI_CRIT_*
    pub strategy: TermsStrategy,
    pub searchable: &'a [(u16, u16)],       // (fid, weight) of searchable_fields_ids / fieldids_weights_map
    pub max_weight: Option<u16>,
    pub authorize_typos: bool,
    pub min_word_len_one_typo: u32,
    pub min_word_len_two_typos: u32,
    pub from: usize,
    pub length: usize,
    pub detailed_scores: bool,
    pub time_budget: Option<std::time::Duration>,
    pub ranking_score_threshold: Option<f64>,
    /// One per `sys::MSI_CRIT_ORDER_BY` entry of `criteria`, in order: the Sort / Asc / Desc rules the shim expanded
    /// (search/new/mod.rs:366-376,640-720).  `RankedScore::Sort { rule, key }` indexes the shim's rank -> value table.
    pub order_keys: &'a [&'a DocOrderKeys],
    /// The distinct field of the request or of the index (`distinct_fid`, search/new/distinct.rs:130-147).
    pub distinct: Option<&'a DocFacetValues>,
    /// One per `sys::MSI_CRIT_GEO_SORT` entry of `criteria`, in order: `AscDesc::{Asc, Desc}(Member::Geo(point))` of the
    /// request (search/new/mod.rs:690-712), with the request's `GeoSortParameter` (documents/geo_sort.rs:12-30).
    pub geo_rules: &'a [GeoRule<'a>],
    pub geo_max_bucket_size: u64,
    pub geo_distance_error_margin: f64,
    /// `exhaustive_number_hits` / `max_total_hits` of `bucket_sort` (bucket_sort.rs:187-191; search/new/mod.rs:894-907)
    pub exhaustive_number_hits: bool,
    pub max_total_hits: Option<usize>,
    /// `GeoSortStrategy` of the request (documents/geo_sort.rs:32-63) as (`sys::MSI_GEO_*`, cache size); the reference's
    /// default is `(sys::MSI_GEO_DYNAMIC, 1000)`.
    pub geo_strategy: (i32, u32),
    /// 0, or — `attributesToSearchOn` — a value naming the restriction the callbacks of `source` answer under (a hash of
    /// the restricted field list): the engine keys what it remembers of stored values by (view, key).  See
    /// `msi_search_params::index_view` in include/msi.h and INTEGRATION.md.
    pub index_view: u64,
}

pub struct GeoRule<'a> { pub points: &'a DocGeoPoints, pub point: [f64; 2], pub ascending: bool }

/// The `_geo` point of every document in HBM (`msi_geo_points`): `[lat, lng]` per docid, `f64::NAN` latitude for the
/// documents outside `geo_faceted_documents_ids`.  Built once per index update from `geo_value`
/// (documents/geo_sort.rs:247-276) over `geo_faceted_documents_ids`.
pub struct DocGeoPoints(NonNull<sys::msi_geo_points>);
unsafe impl Send for DocGeoPoints {}
unsafe impl Sync for DocGeoPoints {}
impl DocGeoPoints {
    pub fn new(ctx: &GpuContext, lat_lng: &[[f64; 2]]) -> Result<Self, GpuError> {
        let mut p = ptr::null_mut();
        check(unsafe { sys::msi_geo_points_create(ctx.0.as_ptr(), lat_lng.as_ptr() as *const f64, lat_lng.len() as u64, &mut p) })?;
        Ok(Self(NonNull::new(p).unwrap()))
    }
}
impl Drop for DocGeoPoints { fn drop(&mut self) { unsafe { sys::msi_geo_points_destroy(self.0.as_ptr()) } } }

/// One u32 order key per document in HBM (`msi_doc_keys`): rank of the first facet value of a field that
/// `ascending_facet_sort` / `descending_facet_sort` meets for the document (sort.rs:95-233), `u32::MAX` without a value.
/// Built once per (index update, field, direction) from `facet_id_f64_docids` + `facet_id_string_docids`.
pub struct DocOrderKeys(NonNull<sys::msi_doc_keys>);
unsafe impl Send for DocOrderKeys {}
unsafe impl Sync for DocOrderKeys {}
impl DocOrderKeys {
    pub fn new(ctx: &GpuContext, keys: &[u32]) -> Result<Self, GpuError> {
        let mut p = ptr::null_mut();
        check(unsafe { sys::msi_doc_keys_create(ctx.0.as_ptr(), keys.as_ptr(), keys.len() as u64, &mut p) })?;
        Ok(Self(NonNull::new(p).unwrap()))
    }
}
impl Drop for DocOrderKeys { fn drop(&mut self) { unsafe { sys::msi_doc_keys_destroy(self.0.as_ptr()) } } }

/// The facet values of the distinct field per document, CSR in HBM (`msi_doc_values`): what `apply_distinct_rule`
/// (search/new/distinct.rs:19-62) reads instead of walking `field_id_docid_facet_f64s` / `_strings` per candidate.
/// `value_ids[offsets[d]..offsets[d + 1]]` = ids (< `n_values`) of the values of document `d`; two documents share a
/// value of the field iff they share an id.  Built once per (index update, field).
pub struct DocFacetValues(NonNull<sys::msi_doc_values>);
unsafe impl Send for DocFacetValues {}
unsafe impl Sync for DocFacetValues {}
impl DocFacetValues {
    pub fn new(ctx: &GpuContext, offsets: &[u64], value_ids: &[u32], n_values: u32) -> Result<Self, GpuError> {
        assert!(!offsets.is_empty() && *offsets.last().unwrap() as usize == value_ids.len());
        let mut p = ptr::null_mut();
        check(unsafe { sys::msi_doc_values_create(ctx.0.as_ptr(), offsets.as_ptr(), value_ids.as_ptr(),
                                                  (offsets.len() - 1) as u64, n_values, &mut p) })?;
        Ok(Self(NonNull::new(p).unwrap()))
    }
}
impl Drop for DocFacetValues { fn drop(&mut self) { unsafe { sys::msi_doc_values_destroy(self.0.as_ptr()) } } }

pub struct RankedOutput {
    pub hits: Vec<(u32, Vec<RankedScore>)>,
    pub candidates: u64,
    pub degraded: bool,
}

type Src<'a> = &'a mut dyn RankingSource;
unsafe fn src<'a>(user: *mut std::ffi::c_void) -> &'a mut Src<'a> { &mut *(user as *mut Src<'a>) }
unsafe fn s<'a>(p: *const u8, n: u32) -> Option<&'a str> { std::str::from_utf8(std::slice::from_raw_parts(p, n as usize)).ok() }
unsafe fn hand(b: Option<&[u8]>, out: *mut *const u8, out_n: *mut usize) -> i32 {
    match b { Some(b) => { *out = b.as_ptr(); *out_n = b.len(); } None => { *out_n = 0; } }
    0
}
unsafe fn keys(v: Vec<u16>, out: *mut u16, cap: u32, n: *mut u32) -> i32 {
    *n = v.len() as u32;
    for (i, k) in v.iter().take(cap as usize).enumerate() { *out.add(i) = *k; }
    0
}
unsafe extern "C" fn r_word(u: *mut std::ffi::c_void, w: *const u8, n: u32, original: i32, o: *mut *const u8, on: *mut usize) -> i32 {
    let Some(w) = s(w, n) else { return -1 }; hand(src(u).word_docids(w, original != 0), o, on)
}
unsafe extern "C" fn r_pair(u: *mut std::ffi::c_void, prox: u32, l: *const u8, ln: u32, r: *const u8, rn: u32, o: *mut *const u8, on: *mut usize) -> i32 {
    let (Some(l), Some(r)) = (s(l, ln), s(r, rn)) else { return -1 }; hand(src(u).word_pair_proximity_docids(prox as u8, l, r), o, on)
}
unsafe extern "C" fn r_exact(u: *mut std::ffi::c_void, w: *const u8, n: u32) -> i32 { s(w, n).map_or(0, |w| src(u).is_exact_word(w) as i32) }
unsafe extern "C" fn r_fid(u: *mut std::ffi::c_void, w: *const u8, n: u32, fid: u32, o: *mut *const u8, on: *mut usize) -> i32 {
    let Some(w) = s(w, n) else { return -1 }; hand(src(u).word_fid_docids(w, fid as u16), o, on)
}
unsafe extern "C" fn r_pos(u: *mut std::ffi::c_void, w: *const u8, n: u32, pos: u32, o: *mut *const u8, on: *mut usize) -> i32 {
    let Some(w) = s(w, n) else { return -1 }; hand(src(u).word_position_docids(w, pos as u16), o, on)
}
unsafe extern "C" fn r_fids(u: *mut std::ffi::c_void, w: *const u8, n: u32, out: *mut u16, cap: u32, cnt: *mut u32) -> i32 {
    let Some(w) = s(w, n) else { return -1 }; keys(src(u).word_fids(w), out, cap, cnt)
}
unsafe extern "C" fn r_positions(u: *mut std::ffi::c_void, w: *const u8, n: u32, out: *mut u16, cap: u32, cnt: *mut u32) -> i32 {
    let Some(w) = s(w, n) else { return -1 }; keys(src(u).word_positions(w), out, cap, cnt)
}
unsafe extern "C" fn r_count(u: *mut std::ffi::c_void, fid: u32, count: u32, o: *mut *const u8, on: *mut usize) -> i32 {
    hand(src(u).field_id_word_count_docids(fid as u16, count as u8), o, on)
}
fn pusher(push: sys::msi_posting_sink, sink: *mut std::ffi::c_void) -> impl FnMut(&[u8]) {
    move |b: &[u8]| unsafe { push(sink, b.as_ptr(), b.len()); }
}
unsafe extern "C" fn r_pfx(u: *mut std::ffi::c_void, p: *const u8, n: u32, original: i32, push: sys::msi_posting_sink, sink: *mut std::ffi::c_void) -> i32 {
    let Some(p) = s(p, n) else { return -1 }; src(u).word_prefix_docids(p, original != 0, &mut pusher(push, sink)) as i32
}
unsafe extern "C" fn r_pfx_fid(u: *mut std::ffi::c_void, p: *const u8, n: u32, fid: u32, push: sys::msi_posting_sink, sink: *mut std::ffi::c_void) -> i32 {
    let Some(p) = s(p, n) else { return -1 }; src(u).word_prefix_fid_docids(p, fid as u16, &mut pusher(push, sink)) as i32
}
unsafe extern "C" fn r_pfx_pos(u: *mut std::ffi::c_void, p: *const u8, n: u32, pos: u32, push: sys::msi_posting_sink, sink: *mut std::ffi::c_void) -> i32 {
    let Some(p) = s(p, n) else { return -1 }; src(u).word_prefix_position_docids(p, pos as u16, &mut pusher(push, sink)) as i32
}
unsafe extern "C" fn r_pfx_pair(u: *mut std::ffi::c_void, prox: u32, w: *const u8, wn: u32, p: *const u8, pn: u32, push: sys::msi_posting_sink, sink: *mut std::ffi::c_void) -> i32 {
    let (Some(w), Some(p)) = (s(w, wn), s(p, pn)) else { return -1 };
    src(u).word_prefix_pair_proximity_docids(prox as u8, w, p, &mut pusher(push, sink)) as i32
}
unsafe extern "C" fn r_pfx_fids(u: *mut std::ffi::c_void, p: *const u8, n: u32, out: *mut u16, cap: u32, cnt: *mut u32) -> i32 {
    let Some(p) = s(p, n) else { return -1 }; keys(src(u).word_prefix_fids(p), out, cap, cnt)
}
unsafe extern "C" fn r_pfx_positions(u: *mut std::ffi::c_void, p: *const u8, n: u32, out: *mut u16, cap: u32, cnt: *mut u32) -> i32 {
    let Some(p) = s(p, n) else { return -1 }; keys(src(u).word_prefix_positions(p), out, cap, cnt)
}
unsafe extern "C" fn r_syn(u: *mut std::ffi::c_void, ws: *const sys::msi_query_token, n: u32, push: sys::msi_synonym_sink, sink: *mut std::ffi::c_void) -> i32 {
    let toks = std::slice::from_raw_parts(ws, n as usize);
    let Some(words) = toks.iter().map(|t| s(t.word, t.len)).collect::<Option<Vec<_>>>() else { return -1 };
    for syn in src(u).synonyms(&words) {
        let t: Vec<_> = syn.iter().map(|w| sys::msi_query_token { word: w.as_ptr(), len: w.len() as u32, is_prefix: 0 }).collect();
        if push(sink, t.as_ptr(), t.len() as u32) < 0 { return -1; }
    }
    0
}

unsafe extern "C" fn r_exact_prefix(u: *mut std::ffi::c_void, p: *const u8, n: u32, push: sys::msi_synonym_sink, sink: *mut std::ffi::c_void) -> i32 {
    let Some(p) = s(p, n) else { return -1 };
    for w in src(u).exact_words_with_prefix(p) {
        let t = sys::msi_query_token { word: w.as_ptr(), len: w.len() as u32, is_prefix: 0 };
        if push(sink, &t, 1) < 0 { return -1; }
    }
    0
}

/// `execute_search` for a keyword query (search/new/mod.rs:808-880) on the device-set engine.
pub fn keyword_search_ranked(dict: &GpuDictionary, sets: &mut GpuDocidSets, source: &mut dyn RankingSource,
                             terms: &[LocatedTerm<'_>], universe: Option<&[u8]>, q: &RankedSearch<'_>)
                             -> Result<RankedOutput, GpuError> {
    let tokens: Vec<Vec<sys::msi_query_token>> = terms.iter().map(|t| t.words.iter().map(|w| match w {
        Some(w) => sys::msi_query_token { word: w.as_ptr(), len: w.len() as u32, is_prefix: (t.is_prefix && !t.is_phrase) as u32 },
        None => sys::msi_query_token { word: ptr::null(), len: 0, is_prefix: 0 },
    }).collect()).collect();
    let located: Vec<_> = terms.iter().zip(&tokens).map(|(t, toks)| sys::msi_located_term {
        words: toks.as_ptr(), n_words: toks.len() as u32,
        is_phrase: (t.is_phrase as u32) | ((t.is_negative as u32) << 1),
        position_start: *t.positions.start() as u32, position_end: *t.positions.end() as u32 }).collect();
    let (fids, weights): (Vec<u16>, Vec<u16>) = q.searchable.iter().copied().unzip();
    let geo: Vec<sys::msi_geo_rule> = q.geo_rules.iter().map(|r| sys::msi_geo_rule {
        points: r.points.0.as_ptr() as *const _, lat: r.point[0], lng: r.point[1], ascending: r.ascending as i32 }).collect();
    let order_ptrs: Vec<*const sys::msi_doc_keys> = q.order_keys.iter().map(|k| k.0.as_ptr() as *const _).collect();
    let params = sys::msi_search_params {
        authorize_typos: q.authorize_typos as u32, min_word_len_one_typo: q.min_word_len_one_typo,
        min_word_len_two_typos: q.min_word_len_two_typos,
        strategy: q.strategy.as_msi(),
        criteria: q.criteria.as_ptr(), n_criteria: q.criteria.len() as u32,
        searchable_fids: fids.as_ptr(), searchable_weights: weights.as_ptr(), n_searchable: fids.len() as u32,
        max_weight: q.max_weight.map_or(-1, |w| w as i32), from: q.from as u32, length: q.length as u32,
        detailed_scores: q.detailed_scores as i32,
        time_budget_us: q.time_budget.map_or(0, |d| d.as_micros().max(1) as u64), stop_after: -1,
        has_score_threshold: q.ranking_score_threshold.is_some() as i32,
        score_threshold: q.ranking_score_threshold.unwrap_or(0.0),
        order_keys: order_ptrs.as_ptr(), n_order_keys: order_ptrs.len() as u32,
        distinct_values: q.distinct.map_or(ptr::null(), |d| d.0.as_ptr() as *const _),
        geo_rules: geo.as_ptr(), n_geo_rules: geo.len() as u32,
        geo_max_bucket_size: q.geo_max_bucket_size.min(u32::MAX as u64) as u32,
        geo_distance_error_margin: q.geo_distance_error_margin,
        exhaustive_number_hits: q.exhaustive_number_hits as i32,
        max_total_hits: q.max_total_hits.map_or(0, |m| m.min(u32::MAX as usize) as u32),
        geo_strategy: q.geo_strategy.0, geo_cache_size: q.geo_strategy.1,
        index_view: q.index_view };
    let mut src_ref: Src<'_> = source;
    let vt = sys::msi_index_vtable { user: &mut src_ref as *mut _ as *mut _, word_docids: Some(r_word),
        word_pair_proximity_docids: Some(r_pair), is_exact_word: Some(r_exact), word_fid_docids: Some(r_fid),
        word_position_docids: Some(r_pos), word_fids: Some(r_fids), word_positions: Some(r_positions),
        field_id_word_count_docids: Some(r_count), word_prefix_docids: Some(r_pfx), word_prefix_fid_docids: Some(r_pfx_fid),
        word_prefix_position_docids: Some(r_pfx_pos), word_prefix_pair_proximity_docids: Some(r_pfx_pair),
        word_prefix_fids: Some(r_pfx_fids), word_prefix_positions: Some(r_pfx_positions), synonyms: Some(r_syn),
        exact_words_with_prefix: Some(r_exact_prefix) };
    let len = q.length.max(1);
    let mut ids = vec![0u32; len];
    let mut details = vec![sys::msi_score_detail::default(); len * sys::MSI_MAX_SCORE_DETAILS];
    let mut n_details = vec![0u32; len];
    let (mut n, mut cand, mut degraded) = (0u32, 0u64, 0i32);
    let (up, ul) = universe.map_or((ptr::null(), 0), |u| (u.as_ptr(), u.len()));
    check(unsafe { sys::msi_keyword_search_ranked(dict.h.as_ptr(), sets.raw(), &vt, located.as_ptr(), located.len() as u32,
                                                  &params, up, ul, ids.as_mut_ptr(), details.as_mut_ptr(),
                                                  n_details.as_mut_ptr(), &mut n, &mut cand, &mut degraded) })?;
    let hits = (0..n as usize).map(|i| {
        let d = &details[i * sys::MSI_MAX_SCORE_DETAILS..][..n_details[i] as usize];
        (ids[i], d.iter().map(|d| match d.kind {
            0 => RankedScore::Words { matching_words: d.a, max_matching_words: d.b },
            1 => RankedScore::Typo { typo_count: d.a, max_typo_count: d.b },
            2 => RankedScore::Proximity { rank: d.a, max_rank: d.b },
            3 => RankedScore::Fid { rank: d.a, max_rank: d.b },
            4 => RankedScore::Position { rank: d.a, max_rank: d.b },
            5 => RankedScore::ExactAttribute { rank: d.a },
            6 => RankedScore::ExactWords { matching_words: d.a, max_matching_words: d.b },
            8 => RankedScore::Sort { rule: d.a, key: d.b },
            9 => RankedScore::GeoSort { rule: d.a, first_docid: d.b },
            _ => RankedScore::Skipped,
        }).collect())
    }).collect();
    Ok(RankedOutput { hits, candidates: cand, degraded: degraded != 0 })
}
