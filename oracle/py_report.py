"""TEST INFRASTRUCTURE — independent pure-Python restatement of the reference's report writer, output_stats()
(src/stats.c:19-298), used only by tests/ to check csrc/report.c byte for byte.  Written from the reference's format
strings, statement by statement, over plain Python containers (dicts / lists), not over the library's structures.

Input `st`: dict with the bs_stats fields as nested lists / dicts:
  snps, indels, multi, dbSNP_sites, dbSNP_var, CpG_ref, CpG_nonref: [all, passed]
  fs, qd, mq: list of [hom, het] (index = value)            (fstats_cts vectors)
  filter_counts: [2][32]; qual: [4][256]; mut, dbsnp_mut: [12][2]
  cov: {coverage: {"all", "var", "CpG": [2], "CpG_inf": [2], "gc": [101]}}   (the gt_cov_stats hash)
  meth: {"ref": [2][101], "nonref": [2][101]}
  read_profile: list of [4] (may be empty), filter_cts[15], filter_bases[15], base_filter[5]
  contigs: list of (name, {"snps": [2], ...}) in header order
"""
MUT = ["A>C", "A>G", "A>T", "C>A", "C>G", "C>T", "G>A", "G>C", "G>T", "T>A", "T>C", "T>G"]  # :20
READ_FILTERS = ["Passed", "Unmapped", "QC_Flags", "SecondaryAlignment", "MateUnmapped", "Duplicate", "NoPosition",
                "NoMatePosition", "MismatchContig", "BadOrientation", "LargeInsertSize", "NoSequence", "LowMAPQ",
                "NotCorrectlyAligned", "PairNotFound"]  # :21-23
BASE_FILTERS = ["Passed", "Trimmed", "Clipped", "Overlapping", "LowQuality"]  # :24-26
FLT_NAME = ["q20", "qd2", "fs60", "mq40"]  # src/init_param.c:15


def output_stats(st, under_conv, over_conv, mapq_thresh, min_qual, date, have_dbsnp):
    w = []
    p = w.append
    p('{\n\t"source": "bs_call_v2.1, under_conversion=%g, over_conversion=%g, mapq_thresh=%d, bq_thread=%d",\n'
      % (under_conv, over_conv, mapq_thresh, min_qual))  # :29
    p('\t"date": "%02d/%02d/%04d",\n' % date)  # :33
    p('\t"filterStats": {\n\t\t"ReadLevel": {\n')  # :34
    p('\t\t\t"%s": {\n\t\t\t\t"Reads": %d,\n\t\t\t\t"Bases": %d\n\t\t\t}' % (READ_FILTERS[0], st["filter_cts"][0], st["filter_bases"][0]))
    for i in range(1, 15):  # :36-40
        if st["filter_cts"][i] > 0:
            p(',\n\t\t\t"%s": {\n\t\t\t\t"Reads": %d,\n\t\t\t\t"Bases": %d\n\t\t\t}' % (READ_FILTERS[i], st["filter_cts"][i], st["filter_bases"][i]))
    p('\n\t\t},\n\t\t"BaseLevel": {\n')  # :41
    p('\t\t\t"%s": %d' % (BASE_FILTERS[0], st["base_filter"][0]))
    for i in range(1, 5):  # :43-47
        if st["base_filter"][i] > 0:
            p(',\n\t\t\t"%s": %d' % (BASE_FILTERS[i], st["base_filter"][i]))
    p('\n\t\t}\n\t},\n\t"totalStats": {\n')  # :48

    def all_passed(key, v):  # :49-57
        p('\t\t"%s": {\n\t\t\t"All": %d,\n\t\t\t"Passed": %d\n\t\t},\n' % (key, v[0], v[1]))

    all_passed("SNPS", st["snps"])
    all_passed("Indels", st["indels"])
    all_passed("Multiallelic", st["multi"])
    if have_dbsnp:
        all_passed("dbSNPSites", st["dbSNP_sites"])
        all_passed("dbSNPVariantSites", st["dbSNP_var"])
    all_passed("RefCpG", st["CpG_ref"])
    all_passed("NonRefCpG", st["CpG_nonref"])
    p('\t\t"QCDistributions": {\n')
    p('\t\t\t"FisherStrand": ')
    term = "{"
    for i, c in enumerate(st["fs"]):  # :61-67
        if c[1] > 0:
            p('%s\n\t\t\t\t"%d": %d' % (term, i, c[1]))
            term = ","
    if term == "{":
        p(term)
    p("\n\t\t\t},\n")
    for key, vec, tail in (("QualityByDepth", st["qd"], "\n\t\t\t},\n"), ("RMSMappingQuality", st["mq"], None)):  # :70-90
        p('\t\t\t"%s": ' % key)
        term = "{"
        for i, c in enumerate(vec):
            if c[0] + c[1] > 0:
                p('%s\n\t\t\t\t"%d": {"NonVariant": %d, "Variant": %d}' % (term, i, c[0], c[1]))
                term = ","
        if term == "{":
            p(term)
        if tail:
            p(tail)
    p('\n\t\t\t}\n\t\t},\t\t"VCFFilterStats": {\n')  # :91
    fc = st["filter_counts"]
    p('\t\t\t"PASS": {"NonVariant": %d, "Variant": %d}' % (fc[0][0], fc[1][0]))
    for i in range(1, 16):  # :93-107
        p(",\n\t\t\t")
        k, f_ix, tmp = i, 0, '"'
        while k:
            if k & 1:
                p("%s%s" % (tmp, FLT_NAME[f_ix]))
                tmp = ","
            k >>= 1
            f_ix += 1
        p('": {"NonVariant": %d, "Variant": %d}' % (fc[0][i], fc[1][i]))
    p("\n\t\t},\n")
    covs = sorted(st["cov"].items())  # HASH_SORT by coverage, :109
    p('\t\t"coverage": {\n')

    def cov_column(get):  # :112-123 and its five repeats
        ix, term = 0, "{"
        for cv, g in covs:
            v = get(g)
            if v != 0:
                if not ix:
                    p("%s\n\t\t\t\t" % term)
                    term = ","
                else:
                    p(", ")
                ix += 1
                p('"%d": %d' % (cv, v))
                ix %= 12

    p('\t\t\t"All": ')
    cov_column(lambda g: g["all"])
    p('\n\t\t\t},\n\t\t\t"Variant": ')
    cov_column(lambda g: g["var"])
    p('\n\t\t\t},\n\t\t\t"RefCpG": ')
    cov_column(lambda g: g["CpG"][0])
    p('\n\t\t\t},\n\t\t\t"RefCpGInf": ')
    cov_column(lambda g: g["CpG_inf"][0])
    p('\n\t\t\t},\n\t\t\t"NonRefCpG": ')
    cov_column(lambda g: g["CpG"][1])
    p('\n\t\t\t},\n\t\t\t"NonRefCpGInf": ')
    cov_column(lambda g: g["CpG_inf"][1])
    p('\n\t\t\t},\n\t\t\t"GC": ')
    term = "{"
    for cv, g in covs:  # :191-201
        if not g["all"]:
            continue
        p('%s\n\t\t\t\t"%d": [\n\t\t\t\t\t' % (term, cv))
        term = ","
        for i in range(100):
            p("%d," % g["gc"][i])
            p("\n\t\t\t\t\t" if (i & 15) == 15 else " ")
        p("%d\n\t\t\t\t]" % g["gc"][100])
    p('\n\t\t\t}\n\t\t},\n\t\t"quality": {\n')
    p('\t\t\t"All": [\n\t\t\t\t')
    for i in range(255):  # :204-207
        p("%d, " % st["qual"][0][i])
        if (i & 15) == 15:
            p("\n\t\t\t\t")
    p("%d\n\t\t\t],\n" % st["qual"][0][255])
    for key, row, tail in (("Variant", 1, "],\n"), ("RefCpG", 2, "],\n"), ("NonRefCpG", 3, "]\n")):  # :209-229
        p('\t\t\t"%s": [\n\t\t\t\t' % key)
        for i in range(255):
            p("%d," % st["qual"][row][i])
            p("\n\t\t\t\t" if (i & 15) == 15 else " ")
        p("%d\n\t\t\t%s" % (st["qual"][row][255], tail))
    p('\t\t},\n\t\t"mutations": {\n')
    for m in range(12):  # :231-238
        p('\t\t\t"%s": { "All": %d, "Passed": %d, "dbSNPAll": %d, "dbSNPPassed": %d }%s\n'
          % (MUT[m], st["mut"][m][0], st["mut"][m][1], st["dbsnp_mut"][m][0], st["dbsnp_mut"][m][1], "," if m < 11 else ""))
    p('\t\t},\n\t\t"methylation": {\n')
    rows = (("AllRefCpg", st["meth"]["ref"][0]), ("PassedRefCpg", st["meth"]["ref"][1]),
            ("AllNonRefCpg", st["meth"]["nonref"][0]), ("PassedNonRefCpg", st["meth"]["nonref"][1]))
    for k, (key, v) in enumerate(rows):  # :240-263
        p('\t\t\t"%s": [\n\t\t\t\t' % key)
        for i in range(100):
            p("%.8g, " % v[i])
            if (i & 15) == 15:
                p("\n\t\t\t\t")
        p("%.8g\n\t\t\t]" % v[100])
        if k < 3:
            p(",\n")
    nr = len(st["read_profile"])
    if nr:  # :264-278
        p(',\n\t\t\t"NonCpGreadProfile": ')
        term = "["
        for i in range(1, nr):
            c = st["read_profile"][i]
            p("%s\n\t\t\t\t[ %d, %d, %d, %d ]" % (term, c[0], c[1], c[2], c[3]))
            term = ","
        p("\n\t\t\t]")
    p('\n\t\t}\n\t},\n\t"contigStats": ')
    term = "{"
    for name, gs in st["contigs"]:  # :281-296
        if gs["snps"][0] == 0:
            continue
        p('%s\n\t\t"%s": {\n' % (term, name))
        term = ","

        def ap(key, v, tail=",\n"):
            p('\t\t\t"%s": {\n\t\t\t\t"All": %d,\n\t\t\t\t"Passed": %d\n\t\t\t}%s' % (key, v[0], v[1], tail))

        ap("SNPS", gs["snps"])
        ap("Indels", gs["indels"])
        ap("Multiallelic", gs["multi"])
        if have_dbsnp:
            ap("dbSNPSites", gs["dbSNP_sites"])
            ap("dbSNPVariantSites", gs["dbSNP_var"])
        ap("RefCpG", gs["CpG_ref"])
        ap("NonRefCpG", gs["CpG_nonref"], "\n\t\t}")
    p("\n\t}\n}\n")
    return "".join(w)
