"""Host-side view of ONE raw get-entries entry, for the rare certificates the host must parse itself
(CTMR_MK_HOST items, serials longer than a record carries).  The batch decode is the GPU's (k_decode_match);
this is RFC 6962 §3.4 / §4.6 framing only, for a single entry the GPU already accepted."""


def certificate_of(leaf_input: bytes, extra_data: bytes) -> bytes:
    """X509Entry of an x509 entry, PrecertChainEntry.pre_certificate of a precert entry (ct-fetch.go:198-204)."""
    entry_type = int.from_bytes(leaf_input[10:12], "big")
    if entry_type == 0:
        n = int.from_bytes(leaf_input[12:15], "big")
        return bytes(leaf_input[15:15 + n])
    n = int.from_bytes(extra_data[0:3], "big")
    return bytes(extra_data[3:3 + n])
