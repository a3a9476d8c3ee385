// ref_json.cpp — TEST INFRASTRUCTURE: the JSON reader the reference vendors (rapidjson, header-only,
// /root/reference/include/rapidjson), called the way src/parsescene.cpp:60-61 calls it (Document::Parse with the default
// flags) and read the way its getFloat3 / getMat4 read numbers (GetDouble, src/parsescene.cpp:20-43).  Compiled where it
// lies by oracle/Makefile into oracle/_ref/libref_json.so; loaded by tests/ only.
#include <rapidjson/include/rapidjson/document.h>

using namespace rapidjson;
#define REF_API extern "C" __attribute__((visibility("default")))

// 1 when Document::Parse accepts the text, else 0
REF_API int ref_json_accepts(const char *text)
{
    Document doc;
    doc.Parse(text);
    return doc.HasParseError() ? 0 : 1;
}

// text: a JSON array of numbers.  Returns how many there are (their GetDouble() in out[0..cap)), -1 on a parse error or a
// document that is not an array of numbers.
REF_API int ref_json_numbers(const char *text, double *out, int cap)
{
    Document doc;
    doc.Parse(text);
    if (doc.HasParseError() || !doc.IsArray()) return -1;
    int n = 0;
    for (Value::ConstValueIterator it = doc.Begin(); it != doc.End(); ++it, ++n) {
        if (!it->IsNumber()) return -1;
        if (n < cap) out[n] = it->GetDouble();
    }
    return n;
}
