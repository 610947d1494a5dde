/* A small command-line parser with the surface the reference's drivers use from their vendored third-party argparse.hpp
 * (FRIES/Ext_Libs/argparse.hpp): `struct MyArgs : public argparse::Args { T &x = kwarg("name", "help").set_default(v); ... };`,
 * optional values as std::shared_ptr<T> &, `argparse::parse<MyArgs>(argc, argv)`, "--name value" on the command line.  Written for
 * this build, not a copy of that library. */
#ifndef FRIES_MINI_ARGPARSE_HPP
#define FRIES_MINI_ARGPARSE_HPP
#include <cstdlib>
#include <iostream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
namespace argparse {
struct Entry {
    std::string name, help, raw, dflt;
    bool has_default = false, optional = false, given = false;
    std::shared_ptr<void> storage;          // the T (or shared_ptr<T>) the reference member is bound to
    void (*assign)(Entry &, const std::string &) = nullptr;
    template <class T> static void assign_to(Entry &e, const std::string &s) { std::istringstream is(s); T v{}; if (!(is >> v)) throw std::runtime_error("invalid value for --" + e.name + ": " + s); *std::static_pointer_cast<T>(e.storage) = v; }
    template <class T> static void assign_opt(Entry &e, const std::string &s) { std::istringstream is(s); T v{}; if (!(is >> v)) throw std::runtime_error("invalid value for --" + e.name + ": " + s); *std::static_pointer_cast<std::shared_ptr<T>>(e.storage) = std::make_shared<T>(v); }
    template <class T> Entry &set_default(const T &v) { std::ostringstream os; os.precision(17); os << v; dflt = os.str(); has_default = true; return *this; }
    template <class T> operator T &() {
        if (!storage) { storage = std::make_shared<T>(); assign = &assign_to<T>; }
        return *std::static_pointer_cast<T>(storage);
    }
    template <class T> operator std::shared_ptr<T> &() {
        if (!storage) { storage = std::make_shared<std::shared_ptr<T>>(); assign = &assign_opt<T>; optional = true; }
        return *std::static_pointer_cast<std::shared_ptr<T>>(storage);
    }
};
template <> inline void Entry::assign_to<std::string>(Entry &e, const std::string &s) { *std::static_pointer_cast<std::string>(e.storage) = s; }
template <> inline void Entry::assign_opt<std::string>(Entry &e, const std::string &s) { *std::static_pointer_cast<std::shared_ptr<std::string>>(e.storage) = std::make_shared<std::string>(s); }
struct Args {
    std::vector<std::shared_ptr<Entry>> entries_;
    Entry &kwarg(const std::string &name, const std::string &help) { entries_.push_back(std::make_shared<Entry>()); entries_.back()->name = name; entries_.back()->help = help; return *entries_.back(); }
    void parse_(int argc, char **argv) {
        std::map<std::string, std::string> kv;
        for (int i = 1; i < argc; i++) {
            std::string a = argv[i];
            if (a == "--help" || a == "-h") { for (auto &e : entries_) std::cout << "  --" << e->name << "  " << e->help << (e->has_default ? " [default " + e->dflt + "]" : "") << "\n"; std::exit(0); }
            if (a.rfind("--", 0) != 0) throw std::runtime_error("expected --option, got " + a);
            std::string key = a.substr(2), val;
            size_t eq = key.find('=');
            if (eq != std::string::npos) { val = key.substr(eq + 1); key = key.substr(0, eq); }
            else { if (i + 1 >= argc) throw std::runtime_error("missing value for --" + key); val = argv[++i]; }
            kv[key] = val;
        }
        for (auto &e : entries_) {
            auto it = kv.find(e->name);
            if (it != kv.end()) { e->assign(*e, it->second); e->given = true; kv.erase(it); }
            else if (e->has_default) e->assign(*e, e->dflt);
            else if (!e->optional) throw std::runtime_error("argument --" + e->name + " is required");
        }
        if (!kv.empty()) throw std::runtime_error("unknown argument --" + kv.begin()->first);
    }
};
template <class T> T parse(int argc, char **argv) {
    T args;
    try { args.parse_(argc, argv); }
    catch (std::exception &ex) { std::cerr << "\nError parsing command line: " << ex.what() << "\n\n"; std::exit(1); }
    return args;
}
}  // namespace argparse
#endif
