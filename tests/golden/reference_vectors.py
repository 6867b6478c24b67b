"""Known-answer vectors transcribed from the reference's own unit tests (the only golden data the
reference holds for the decode path).  Each entry cites the reference test it comes from.

Used twice: tests/test_oracle_golden.py pins the CPU oracle with them (CPU), and
tests/test_gpu_parity.py runs the same lines through the HIP path (GPU).
"""

RFC5424, LTSV, GELF, RFC3164 = 0, 1, 2, 3

S = lambda v: ("String", v)  # noqa: E731

LTSV_SCHEMA = {"input": {"ltsv_schema": {"counter": "u64", "score": "i64", "mean": "f64", "done": "bool"}}}
LTSV_SUFFIX_1 = {"input": {"ltsv_schema": {"counter": "U64", "score": "I64", "mean": "f64", "done": "bool"},
                           "ltsv_suffixes": {"u64": "_u64", "i64": "_i64", "F64": "_f64", "Bool": "_bool"}}}
LTSV_SUFFIX_2 = {"input": {"ltsv_schema": {"counter_u64": "U64", "score_i64": "I64", "mean_f64": "f64",
                                           "done_bool": "bool"},
                           "ltsv_suffixes": {"u64": "_u64", "i64": "_i64", "f64": "_f64", "bool": "_bool"}}}

# Each vector: dict(fmt, line, config, ok=dict(field -> expected) | err="message", src="file:line")
# `pairs` lists (sd_index, key, (kind, value)) that must be present; `n_sd` the number of elements.
VECTORS = [
    dict(
        src="rfc5424_decoder.rs:244-278 test_rfc5424", fmt=RFC5424, config=None,
        line=r'<23>1 2015-08-05T15:53:45.637824Z testhostname appname 69 42 [origin@123 software="te\st sc\"ript" swVersion="0.0.1"] test message',
        ok=dict(facility=2, severity=7, ts=1438790025.637824, hostname="testhostname", appname="appname",
                procid="69", msgid="42", msg="test message", n_sd=1, sd_ids=["origin@123"],
                pairs=[(0, "_software", S('te\\st sc"ript')), (0, "_swVersion", S("0.0.1"))]),
    ),
    dict(
        src="rfc5424_decoder.rs:280-314 test_rfc5424_multiple_sd", fmt=RFC5424, config=None,
        line=r'<23>1 2015-08-05T15:53:45.637824Z testhostname appname 69 42 [origin@123 software="te\st sc\"ript" swVersion="0.0.1"][master@456 key="value" key2="value2"] test message',
        ok=dict(facility=2, severity=7, ts=1438790025.637824, hostname="testhostname", appname="appname",
                procid="69", msgid="42", msg="test message", n_sd=2, sd_ids=["origin@123", "master@456"],
                pairs=[(0, "_software", S('te\\st sc"ript')), (0, "_swVersion", S("0.0.1")),
                       (1, "_key", S("value")), (1, "_key2", S("value2"))]),
    ),
    dict(
        src="ltsv_decoder.rs:269-316 test_ltsv_suffixes", fmt=LTSV, config=LTSV_SUFFIX_1,
        line="time:[10/Oct/2000:13:55:36 -0700]\tdone:true\tscore:-1\tmean:0.42\tcounter:42\tlevel:3\thost:testhostname\tname1:value1\tname 2: value 2\tn3:v3\tmessage:this is a test",
        ok=dict(n_sd=1, pairs=[(0, "_counter_u64", ("U64", 42)), (0, "_score_i64", ("I64", -1)),
                               (0, "_mean_f64", ("F64", 0.42)), (0, "_done_bool", ("Bool", True))]),
    ),
    dict(
        src="ltsv_decoder.rs:318-366 test_ltsv_suffixes_2", fmt=LTSV, config=LTSV_SUFFIX_2,
        line="time:[10/Oct/2000:13:55:36 -0700]\tdone_bool:true\tscore_i64:-1\tmean_f64:0.42\tcounter_u64:42\tlevel:3\thost:testhostname\tname1:value1\tname 2: value 2\tn3:v3\tmessage:this is a test",
        ok=dict(n_sd=1, pairs=[(0, "_counter_u64", ("U64", 42)), (0, "_score_i64", ("I64", -1)),
                               (0, "_mean_f64", ("F64", 0.42)), (0, "_done_bool", ("Bool", True))]),
    ),
    dict(
        src="ltsv_decoder.rs:368-379 test_ltsv", fmt=LTSV, config=LTSV_SCHEMA,
        line="time:1438790025.99\thost:testhostname\tname1:value1\tname 2: value 2\tn3:v3",
        ok=dict(ts=1438790025.99),
    ),
    dict(
        src="ltsv_decoder.rs:381-393 test_ltsv2", fmt=LTSV, config=LTSV_SCHEMA,
        line="time:[2015-08-05T15:53:45.637824Z]\thost:testhostname\tname1:value1\tname 2: value 2\tn3:v3",
        ok=dict(ts=1438790025.637824),
    ),
    dict(
        src="ltsv_decoder.rs:395-472 test_ltsv_3", fmt=LTSV, config=LTSV_SCHEMA,
        line="time:[10/Oct/2000:13:55:36.3 -0700]\tdone:true\tscore:-1\tmean:0.42\tcounter:42\tlevel:3\thost:testhostname\tname1:value1\tname 2: value 2\tn3:v3\tmessage:this is a test",
        ok=dict(ts=971211336.3, severity=3, hostname="testhostname", msg="this is a test", n_sd=1,
                pairs=[(0, "_name1", S("value1")), (0, "_name 2", S(" value 2")), (0, "_n3", S("v3")),
                       (0, "_counter", ("U64", 42)), (0, "_score", ("I64", -1)), (0, "_mean", ("F64", 0.42)),
                       (0, "_done", ("Bool", True))]),
    ),
    dict(
        src="ltsv_decoder.rs:474-487 test_ltsv4", fmt=LTSV, config=LTSV_SCHEMA,
        line="time:[5/Aug/2015:15:53:45.637824 -0000]\thost:testhostname\tname1:value1\tname 2: value 2\tn3:v3",
        ok=dict(ts=1438790025.637824),
    ),
    dict(
        src="gelf_decoder.rs:133-170 test_gelf_decoder", fmt=GELF, config=None,
        line=r'{"version":"1.1", "host": "example.org","short_message": "A short message that helps you identify what is going on", "full_message": "Backtrace here\n\nmore stuff", "timestamp": 1385053862.3072, "level": 1, "_user_id": 9001, "_some_info": "foo", "_some_env_var": "bar"}',
        ok=dict(ts=1385053862.3072, hostname="example.org",
                msg="A short message that helps you identify what is going on",
                full_msg="Backtrace here\n\nmore stuff", severity=1, n_sd=1,
                pairs=[(0, "_user_id", ("U64", 9001)), (0, "_some_info", S("foo")), (0, "_some_env_var", S("bar"))]),
    ),
    dict(src="gelf_decoder.rs:172-177 test_gelf_decoder_bad_key", fmt=GELF, config=None,
         line='{"some_key": []}', err="Invalid value type in structured data"),
    dict(src="gelf_decoder.rs:179-184 test_gelf_decoder_bad_timestamp", fmt=GELF, config=None,
         line='{"timestamp": "a string not a timestamp", "host": "anhostname"}', err="Invalid GELF timestamp"),
    dict(src="gelf_decoder.rs:186-190 test_gelf_decoder_invalid_input", fmt=GELF, config=None,
         line='{some_key = "some_value"}', err="Invalid GELF input, unable to parse as a JSON object"),
    dict(src="gelf_decoder.rs:192-197 test_gelf_decoder_wrong_version", fmt=GELF, config=None,
         line='{"version":"42"}', err="Unsupported GELF version"),
    dict(src="gelf_decoder.rs:199-205 test_gelf_decoder_severity_to_high", fmt=GELF, config=None,
         line='{"level": 8}', err="Invalid severity level (too high)"),
]

# Behaviour read off the reference source line by line; not reference tests,
# but unambiguous consequences of rfc5424_decoder.rs that oracle and GPU path must agree on.
RFC5424_HDR = "<13>1 2015-08-05T15:53:45Z h a p m "
DERIVED_RFC5424 = [
    ("", "Unsupported BOM"),
    ("hello", "Unsupported BOM"),
    ("\ufeff", "The priority should be inside brackets"),
    ("\ufeffx<13>1", "The priority should be inside brackets"),
    ("<>1 x", "Invalid priority"),
    ("<256>1 x", "Invalid priority"),
    ("<-1>1 x", "Invalid priority"),
    ("<1x>1 x", "Invalid priority"),
    ("<+>1 x", "Invalid priority"),
    ("<13", "Missing version"),
    ("<13 1", "Missing version"),
    ("<13>2 x", "Unsupported version"),
    ("<13>1> x", "Unsupported version"),
    ("<13> x", "Unsupported version"),
    ("<13>1", "Missing timestamp"),
    ("<13>1 ", "Unable to parse the date from RFC3339 to Unix time in RFC5424 decoder"),
    ("<13>1 - h a p m - x", "Unable to parse the date from RFC3339 to Unix time in RFC5424 decoder"),
    ("<13>1 2015-08-05T15:53:45Z", "Missing hostname"),
    ("<13>1 2015-08-05T15:53:45Z h", "Missing application name"),
    ("<13>1 2015-08-05T15:53:45Z h a", "Missing process id"),
    ("<13>1 2015-08-05T15:53:45Z h a p", "Missing message id"),
    ("<13>1 2015-08-05T15:53:45Z h a p m", "Missing message data"),
    (RFC5424_HDR, "Missing log message"),
    (RFC5424_HDR + "x", "Malformated RFC5424 message"),
    (RFC5424_HDR + "[id]", "Missing structured data"),
    (RFC5424_HDR + "[id] m", "Missing ] after structured data"),
    (RFC5424_HDR + '[a b="c"]', "Missing log message"),
    (RFC5424_HDR + '[a b="c"]x', "Malformated RFC5424 message"),
    (RFC5424_HDR + '[a b= "c"] m', "Format error in the structured data"),
    (RFC5424_HDR + '[a b="c" =] m', "Format error in the structured data"),
    (RFC5424_HDR + '[a b="c', "Missing ] after structured data"),
    (RFC5424_HDR + '[a é="c"] m', "Format error in the structured data"),
]


# ---- RFC3164 decoder (rfc3164_decoder.rs:215-425).  The reference's expected timestamps use the CURRENT year
# (ts_from_partial_date_time); the decoder under test is configured with RFC3164_YEAR instead of reading the clock.
import calendar as _cal  # noqa: E402

RFC3164_YEAR = 2026
RFC3164_CONFIG = {"rfc3164": {"current_year": RFC3164_YEAR}}
_ts = lambda y, mo, d, h, mi, s: float(_cal.timegm((y, mo, d, h, mi, s)))  # noqa: E731
_M3164 = r'appname 69 42 [origin@123 software="te\st sc\"ript" swVersion="0.0.1"] test message'
RFC3164_VECTORS = [
    dict(src="rfc3164_decoder.rs:222-242 test_rfc3164_decode_nopri", fmt=RFC3164, config=RFC3164_CONFIG,
         line="Aug  6 11:15:24 testhostname " + _M3164,
         ok=dict(facility=None, severity=None, ts=_ts(RFC3164_YEAR, 8, 6, 11, 15, 24), hostname="testhostname", appname=None, procid=None,
                 msgid=None, msg=_M3164, full_msg="Aug  6 11:15:24 testhostname " + _M3164, sd_none=True)),
    dict(src="rfc3164_decoder.rs:244-264 test_rfc3164_decode_with_pri", fmt=RFC3164, config=RFC3164_CONFIG,
         line="<13>Aug  6 11:15:24 testhostname " + _M3164,
         ok=dict(facility=1, severity=5, ts=_ts(RFC3164_YEAR, 8, 6, 11, 15, 24), hostname="testhostname", msg=_M3164,
                 full_msg="<13>Aug  6 11:15:24 testhostname " + _M3164, sd_none=True)),
    dict(src="rfc3164_decoder.rs:266-286 test_rfc3164_decode_with_pri_year", fmt=RFC3164, config=RFC3164_CONFIG,
         line="<13>2020 Aug  6 11:15:24 testhostname " + _M3164,
         ok=dict(facility=1, severity=5, ts=_ts(2020, 8, 6, 11, 15, 24), hostname="testhostname", msg=_M3164,
                 full_msg="<13>2020 Aug  6 11:15:24 testhostname " + _M3164, sd_none=True)),
    dict(src="rfc3164_decoder.rs:288-308 test_rfc3164_decode_with_pri_year_tz", fmt=RFC3164, config=RFC3164_CONFIG,
         line="<13>2020 Aug 6 05:15:24 America/Sao_Paulo testhostname " + _M3164,
         ok=dict(facility=1, severity=5, ts=_ts(2020, 8, 6, 8, 15, 24), hostname="testhostname", msg=_M3164,
                 full_msg="<13>2020 Aug 6 05:15:24 America/Sao_Paulo testhostname " + _M3164, sd_none=True)),
    dict(src="rfc3164_decoder.rs:310-330 test_rfc3164_decode_tz_no_year", fmt=RFC3164, config=RFC3164_CONFIG,
         line="Aug  6 11:15:24 UTC testhostname " + _M3164,
         ok=dict(facility=None, severity=None, ts=_ts(RFC3164_YEAR, 8, 6, 11, 15, 24), hostname="testhostname", msg=_M3164,
                 full_msg="Aug  6 11:15:24 UTC testhostname " + _M3164, sd_none=True)),
    dict(src="rfc3164_decoder.rs:332-340 test_rfc3164_decode_invalid_event", fmt=RFC3164, config=RFC3164_CONFIG,
         line="test message", err="Malformed RFC3164 event: Invalid timestamp or hostname"),  # the test only asserts is_err()
    dict(src="rfc3164_decoder.rs:342-350 test_rfc3164_decode_invalid_date", fmt=RFC3164, config=RFC3164_CONFIG,
         line="Aug  36 11:15:24 testhostname " + _M3164, err="Malformed RFC3164 event: Invalid timestamp or hostname"),
    dict(src="rfc3164_decoder.rs:352-375 test_rfc3164_decode_custom_with_year", fmt=RFC3164, config=RFC3164_CONFIG,
         line="testhostname: 2020 Aug  6 11:15:24 UTC: appname 69 42 some test message",
         ok=dict(facility=None, severity=None, ts=_ts(2020, 8, 6, 11, 15, 24), hostname="testhostname", msg="appname 69 42 some test message",
                 full_msg="testhostname: 2020 Aug  6 11:15:24 UTC: appname 69 42 some test message", sd_none=True)),
    dict(src="rfc3164_decoder.rs:377-397 test_rfc3164_decode_custom_with_year_notz", fmt=RFC3164, config=RFC3164_CONFIG,
         line="testhostname: 2019 Mar 27 12:09:39: appname: a test message",
         ok=dict(ts=_ts(2019, 3, 27, 12, 9, 39), hostname="testhostname", msg="appname: a test message",
                 full_msg="testhostname: 2019 Mar 27 12:09:39: appname: a test message", sd_none=True)),
    dict(src="rfc3164_decoder.rs:399-419 test_rfc3164_decode_custom_with_pri", fmt=RFC3164, config=RFC3164_CONFIG,
         line="<13>testhostname: 2019 Mar 27 12:09:39 UTC: appname: test message",
         ok=dict(facility=1, severity=5, ts=_ts(2019, 3, 27, 12, 9, 39), hostname="testhostname", msg="appname: test message",
                 full_msg="<13>testhostname: 2019 Mar 27 12:09:39 UTC: appname: test message", sd_none=True)),
    dict(src="rfc3164_decoder.rs:421-441 test_rfc3164_decode_custom_trimed", fmt=RFC3164, config=RFC3164_CONFIG,
         line="<13>testhostname: 2019 Mar 27 12:09:39 UTC: appname: test message \n",
         ok=dict(facility=1, severity=5, ts=_ts(2019, 3, 27, 12, 9, 39), hostname="testhostname",
                 full_msg="<13>testhostname: 2019 Mar 27 12:09:39 UTC: appname: test message", sd_none=True)),
]
