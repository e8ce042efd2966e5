// Stand-in for <boost/program_options.hpp> (build check only, see ../README.md): the subset apps/m17-demod.cpp:386-443 uses —
// bool switches with "long,s" names, --help / --version presence, store / notify.
#pragma once
#include <map>
#include <ostream>
#include <string>
#include <vector>
namespace boost { namespace program_options {

struct value_semantic { bool* target; };
inline value_semantic* bool_switch(bool* target) { *target = false; return new value_semantic{target}; }

struct option_entry { std::string long_name; char short_name; value_semantic* value; std::string text; };

class options_description
{
public:
    std::string caption;
    std::vector<option_entry> entries;
    struct adder {
        options_description* owner;
        adder& operator()(const char* name, const char* text) { return (*this)(name, nullptr, text); }
        adder& operator()(const char* name, value_semantic* v, const char* text)
        {
            std::string n(name);
            const auto comma = n.find(',');
            owner->entries.push_back({n.substr(0, comma), comma == std::string::npos ? char(0) : n[comma + 1], v, text});
            return *this;
        }
    };
    explicit options_description(const char* c) : caption(c) {}
    adder add_options() { return adder{this}; }
};
inline std::ostream& operator<<(std::ostream& os, const options_description& d)
{
    os << d.caption << ":\n";
    for (const auto& e : d.entries) os << "  --" << e.long_name << "  " << e.text << "\n";
    return os;
}

class variables_map : public std::map<std::string, int>
{
public:
    size_t count(const std::string& key) const { return std::map<std::string, int>::count(key); }
};
struct parsed_options { std::vector<const option_entry*> seen; };
inline parsed_options parse_command_line(int argc, char* argv[], const options_description& d)
{
    parsed_options p;
    for (int i = 1; i < argc; ++i) {
        const std::string a(argv[i]);
        for (const auto& e : d.entries)
            if (a == "--" + e.long_name || (e.short_name && a == std::string("-") + e.short_name)) p.seen.push_back(&e);
    }
    return p;
}
inline void store(const parsed_options& p, variables_map& vm)
{
    for (const auto* e : p.seen) { vm[e->long_name] = 1; if (e->value) *e->value->target = true; }
}
inline void notify(variables_map&) {}

}} // boost::program_options
